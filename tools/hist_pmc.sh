#!/bin/bash
# PMC passes + steady-state kernel trace of the histogram kernels at configs[1] (tools/hist_probe.py).
# usage (GPU box, repo root): bash tools/hist_pmc.sh <outdir>     (env HG_HIST_* selects the variant)
set -u
OUT=$1
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() { local name=$1; shift; (cd /tmp && HG_HIST_ITERS=6 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o p -- python "$ROOT/tools/hist_probe.py" > "$ROOT/$OUT/$name.log" 2>&1); }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES
run occ SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python tools/pmc_summary.py "$OUT" k_hist > "$OUT/hist_pmc.txt" 2>&1
(cd /tmp && HG_HIST_ITERS=40 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -o t -- python "$ROOT/tools/hist_probe.py" > "$ROOT/$OUT/trace.log" 2>&1)
find "$OUT" -name "*.csv" -size +300k -delete
cat "$OUT/hist_pmc.txt"; cat "$OUT/trace.log" | tail -2
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
