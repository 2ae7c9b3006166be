#!/bin/bash
# the same with the HIP runtime's hardware-queue count capped (GPU_MAX_HW_QUEUES): does the anomaly follow the number of queues two processes put on one GPU?
set -u
TAG=${1:-r05ddpab2}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for q in 2 4; do
  GPU_MAX_HW_QUEUES=$q HG_DIST_BACKEND=gloo HG_G_OVERLAP_DDP=1 timeout 150 python bench.py --gpus 2 --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_ov1_q$q.json 2> $OUT/err_q$q.txt
done
for f in $OUT/bench_ov*.json; do python -c "
import json
L=[l for l in open('$f') if l.startswith('{')]
if L:
    d=json.loads(L[-1]); print('$f'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],1), (d.get('ddp') or {}).get('allreduce_ms_per_step'))
else: print('$f'.split('/')[-1], 'no line')"; done
