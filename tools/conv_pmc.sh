#!/bin/bash
# PMC passes of the convolution kernels, one launch set per tile shape (tools/conv_pmc.py).
# usage (GPU box, repo root): bash tools/conv_pmc.sh <outdir>
set -u
OUT=$1
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() { local name=$1; shift; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o p -- python "$ROOT/tools/conv_pmc.py" > "$ROOT/$OUT/$name.log" 2>&1); }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES
run occ SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python tools/pmc_summary.py "$OUT" "k_conv<" "k_wgrad<" > "$OUT/conv_pmc.txt" 2>&1
find "$OUT" -name "*.csv" -size +300k -delete
cat "$OUT/conv_pmc.txt"
