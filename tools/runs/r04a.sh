#!/bin/bash
# GPU run r04a: the new parity tests of round 4 (active-hinge plain step, reference goldens at C3 width, launcher, flush) and
# the PMC pass on stall reasons of the dense histogram kernels.
set -u
ROOT=$(pwd); OUT=gpurun_out/r04a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x -p no:cacheprovider \
  "tests/test_c3_parity_gpu.py::test_c3_train_step_matches_oracle" "tests/test_c3_parity_gpu.py::test_c3_networks_match_reference_golden" \
  tests/test_nets_gpu.py::test_plain_step_d_phase_matches_oracle tests/test_nets_gpu.py::test_train_step_matches_oracle \
  tests/test_trainer_io_gpu.py tests/test_bench_gpu.py tests/test_ddp_step_gpu.py > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/c3_parity.json 2>/dev/null
run() { local name=$1; shift; (cd /tmp && HG_HIST_ITERS=6 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o p -- python "$ROOT/tools/hist_probe.py" > "$ROOT/$OUT/$name.log" 2>&1); }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA
run p2 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VALU
run p3 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
run p4 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM
python tools/pmc_summary.py "$OUT" k_hist > "$OUT/hist_pmc.txt" 2>&1
find "$OUT" -name "*.csv" -size +300k -delete
cat "$OUT/hist_pmc.txt" | head -120
for l in p1 p2 p3 p4; do tail -1 $OUT/$l.log; done
