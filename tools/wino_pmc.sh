#!/bin/bash
# PMC passes + an un-profiled kernel trace of the Winograd kernels (tools/wino_pmc.py).
# usage (GPU box, repo root): bash tools/wino_pmc.sh <outdir>
set -u
OUT=$1
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() { local name=$1; shift; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o p -- python "$ROOT/tools/wino_pmc.py" > "$ROOT/$OUT/$name.log" 2>&1); }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES
run occ SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
python tools/pmc_summary.py "$OUT" "k_wino" > "$OUT/wino_pmc.txt" 2>&1
# per-launch traffic of bench.py's roofline launch alone (256 -> 128 channels, 64x64, batch 32): separate FETCH / WRITE passes
runr() { local name=$1; shift; (cd /tmp && HG_PMC_ONLY=roofline timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/roof/$name" -o p -- python "$ROOT/tools/wino_pmc.py" > "$ROOT/$OUT/roof_$name.log" 2>&1); }
mkdir -p "$OUT/roof"
runr fetch FETCH_SIZE
runr write WRITE_SIZE
python tools/pmc_summary.py "$OUT/roof" "k_wino" > "$OUT/wino_roofline_traffic.txt" 2>&1
(cd /tmp && HG_PMC_ONLY=roofline HG_ONE_ITERS=600 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/roof/trace" -o t -- python "$ROOT/tools/wino_pmc.py" > "$ROOT/$OUT/roof_trace.log" 2>&1)
cp "$OUT"/roof/trace/*kernel_stats.csv "$OUT/wino_roofline_kernel_stats.csv" 2>/dev/null
(cd /tmp && HG_PMC_ONLY=leading HG_ONE_ITERS=600 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/lead/trace" -o t -- python "$ROOT/tools/wino_pmc.py" > "$ROOT/$OUT/lead_trace.log" 2>&1)
cp "$OUT"/lead/trace/*kernel_stats.csv "$OUT/wino_leading_kernel_stats.csv" 2>/dev/null
(cd /tmp && HG_ONE_ITERS=40 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -o t -- python "$ROOT/tools/wino_pmc.py" > "$ROOT/$OUT/trace.log" 2>&1)
find "$OUT" -name "*.csv" -size +300k -delete
cat "$OUT/wino_pmc.txt"
head -30 "$OUT"/trace/*kernel_stats.csv 2>/dev/null
