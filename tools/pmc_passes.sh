#!/bin/bash
# PMC collection for the hist workload: separate passes (counters only with --kernel-trace), csv output.
# usage (on the GPU box, from repo root): bash tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
ROOT=$(pwd)
run() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o p -- \
     python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "${BENCH_ARGS[@]}" > "$ROOT/$OUT/$name.log" 2>&1)
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
find "$OUT" -name "*.csv" | head -20
