#!/bin/bash
# Trace of the two-gloo-ranks-on-one-GPU anomaly (profiles/r04_ddp_gloo_knobs.json): HG_G_OVERLAP_DDP=1 vs 0, HIP API + kernel + copy traces
set -u
TAG=${1:-r05ddp}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for x in 1 0; do
  (cd /tmp && HG_DIST_BACKEND=gloo HG_G_OVERLAP_DDP=$x timeout 280 rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $OUT/ov$x -- python $ROOT/bench.py --gpus 2 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_ov$x.json 2> $OUT/bench_ov$x.err)
  grep -o '"ms_per_step": [0-9.]*' $OUT/bench_ov$x.json | head -1
  python tools/hip_api_summary.py $OUT/ov$x 10 > $OUT/summary_ov$x.txt 2>&1
  head -60 $OUT/summary_ov$x.txt | cut -c1-170
  find $OUT/ov$x -name "*.csv" -size +200k -delete
done
