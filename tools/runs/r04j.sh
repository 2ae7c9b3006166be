#!/bin/bash
# GPU run r04j: why is the two-rank gloo-on-one-GPU step 2.2 s (round 2: 0.295 s)?  Knob matrix.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04j; mkdir -p $OUT
run() { local name=$1; shift; env "$@" HG_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 6 --warmup 3 --no-roofline > $OUT/$name.json 2> $OUT/$name.err; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.strip()][-1]); print(sys.argv[2], round(d['ms_per_step'],1), 'ms/step', 'allreduce', round(d['ddp']['allreduce_ms_per_step'],1))
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
run default A=1
run no_overlap HG_G_OVERLAP=0
run one_bucket HG_DDP_BUCKETS=1
run blocking_stats HG_LAZY_STATS=0
run no_overlap_one_bucket HG_G_OVERLAP=0 HG_DDP_BUCKETS=1
run no_wgrad_stream HG_WGRAD_STREAM=0
