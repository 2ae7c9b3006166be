#!/bin/bash
# un-traced A/B of the second-stream generator forward under data parallelism, two gloo ranks sharing one GPU (cf. profiles/r04_ddp_gloo_knobs.json)
set -u
TAG=${1:-r05ddpab}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for x in 1 0 1 0; do
  HG_DIST_BACKEND=gloo HG_G_OVERLAP_DDP=$x timeout 200 python bench.py --gpus 2 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_ov${x}_$RANDOM.json 2> $OUT/err.txt
done
for f in $OUT/bench_ov*.json; do python -c "
import json
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],1), (d.get('ddp') or {}).get('allreduce_ms_per_step'))"; done
