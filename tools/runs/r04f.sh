#!/bin/bash
# GPU run r04f: kernel-trace durations of k_hist_bwd with / without the shared-reciprocal K loop (same box, interleaved)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04f; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for v in 0 1; do
  (cd /tmp && HG_BWD_SHARE_RCP=$v HG_FWD_SHARE_RCP=$v HG_HIST_ITERS=40 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t${v}_$rep" -o t -- python "$ROOT/tools/hist_probe.py" > "$OUT/t${v}_$rep.log" 2>&1)
  f=$(find "$OUT/t${v}_$rep" -name "*kernel_stats.csv" | head -1)
  echo "share=$v rep=$rep"; grep "k_hist_bwd\|k_hist_fwd" "$f" | awk -F'","' '{n=split($1,a,"<"); printf "  %s calls %s avg %.1f us min %.1f us\n", substr($1,1,60), $2, $4/1000, $6/1000}'
done; done | tee $OUT/summary.txt
find "$OUT" -name "*.csv" -size +100k -delete
