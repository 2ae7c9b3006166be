#!/bin/bash
# GPU run r04e: shared-reciprocal k_hist_bwd (parity + A/B), the fixed parity tests, the bf16x6 suite again
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04e; mkdir -p $OUT
export TMPDIR=/tmp
P="python -m pytest -m gpu -q -p no:cacheprovider"
timeout 600 $P -x tests/test_hist_gpu.py tests/test_hist_big_gpu.py tests/test_hist_planes_gpu.py > $OUT/pytest_hist.log 2>&1; tail -3 $OUT/pytest_hist.log
for i in 1 2; do
  HG_BWD_SHARE_RCP=0 HG_HIST_ITERS=40 timeout 100 python tools/hist_probe.py 2>&1 | tail -1 | sed 's/^/bwd share=0: /'
  HG_BWD_SHARE_RCP=1 HG_HIST_ITERS=40 timeout 100 python tools/hist_probe.py 2>&1 | tail -1 | sed 's/^/bwd share=1: /'
done | tee $OUT/ab.txt
HG_HIST_INSZ=150 HG_HIST_ITERS=40 timeout 100 python tools/hist_probe.py 2>&1 | tail -1 | tee -a $OUT/ab.txt
timeout 900 $P tests/test_c3_parity_gpu.py::test_c3_networks_match_reference_golden "tests/test_c3_parity_gpu.py::test_c3_train_step_matches_oracle" tests/test_nets_gpu.py::test_plain_step_d_phase_matches_oracle > $OUT/pytest_fixed.log 2>&1; tail -4 $OUT/pytest_fixed.log
cp gpurun_out/c3_parity.json $OUT/c3_parity_f32.json 2>/dev/null
rm -f gpurun_out/c3_parity.json
HG_CONV_PRECISION=b6 timeout 600 $P --tb=line tests/test_c3_parity_gpu.py tests/test_nets_gpu.py > $OUT/pytest_b6.log 2>&1; tail -6 $OUT/pytest_b6.log
cp gpurun_out/c3_parity.json $OUT/c3_parity_b6.json 2>/dev/null
