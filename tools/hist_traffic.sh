#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, counters only with --kernel-trace) over tools/hist_probe.py
# usage: bash tools/hist_traffic.sh <outdir>   (env HG_HIST_* selects the variant)
set -u
OUT=$1
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && HG_HIST_ITERS=6 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/$c" -o p -- python "$ROOT/tools/hist_probe.py" > "$ROOT/$OUT/$c.log" 2>&1)
done
python tools/pmc_summary.py "$OUT" k_hist k_thr > "$OUT/traffic.txt" 2>&1
find "$OUT" -name "*.csv" -size +300k -delete
cat "$OUT/traffic.txt"
