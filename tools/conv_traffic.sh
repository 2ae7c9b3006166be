#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, counters only with --kernel-trace) over tools/conv_pmc_one.py: bench.py's
# roofline launch (k_conv forward 256->128 ch, 64x64, batch 32), k_wgrad of the same layer and the histogram kernels at configs[1].
# usage (GPU box, repo root): bash tools/conv_traffic.sh <outdir>
set -u
OUT=$1
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do   # (HG_WINO=0: the DIRECT kernels of the roofline layer -- Winograd has its own passes, tools/wino_pmc.sh)
  (cd /tmp && HG_WINO=0 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/$c" -o p -- python "$ROOT/tools/conv_pmc_one.py" > "$ROOT/$OUT/$c.log" 2>&1)
done
python tools/pmc_summary.py "$OUT" "k_conv<" "k_wgrad<" k_hist > "$OUT/traffic.txt" 2>&1
(cd /tmp && HG_WINO=0 HG_ONE_ITERS=60 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -o t -- python "$ROOT/tools/conv_pmc_one.py" > "$ROOT/$OUT/trace.log" 2>&1)
find "$OUT" -name "*.csv" -size +300k -delete
cat "$OUT/traffic.txt"
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -r head -8
