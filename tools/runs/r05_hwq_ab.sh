#!/bin/bash
# single-process step with the HIP runtime's hardware-queue cap raised (the step uses ~7 streams on 4 queues by default)
set -u
TAG=${1:-r05hwq}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for q in 2 3 4 2 3 4; do
  GPU_MAX_HW_QUEUES=$q timeout 120 python bench.py --steps 32 --warmup 6 --no-roofline --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_q${q}_$RANDOM.json 2> $OUT/err.txt
done
for f in $OUT/bench_q*.json; do python -c "
import json
L=[l for l in open('$f') if l.startswith('{')]
d=json.loads(L[-1]); print('$f'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],2))"; done
