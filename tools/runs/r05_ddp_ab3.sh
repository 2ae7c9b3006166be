#!/bin/bash
# the data-parallel step as ONE rank sees it on a node (RCCL, world size 1 forced): second-stream generator forward on / off
set -u
TAG=${1:-r05ddpab3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for x in 1 0 1 0; do
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + RANDOM % 500)) HG_DIST_BACKEND=nccl HG_DIST_FORCE=1 HG_G_OVERLAP_DDP=$x \
    timeout 150 python bench.py --gpus 1 --steps 32 --warmup 6 --no-roofline --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_ov${x}_$RANDOM.json 2> $OUT/err.txt
done
for f in $OUT/bench_ov*.json; do python -c "
import json
L=[l for l in open('$f') if l.startswith('{')]
if L:
    d=json.loads(L[-1]); print('$f'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],2), (d.get('ddp') or {}).get('backend'), (d.get('ddp') or {}).get('allreduce_ms_per_step'))
else: print('$f'.split('/')[-1], 'no line')"; done; tail -3 $OUT/err.txt
