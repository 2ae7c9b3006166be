#!/bin/bash
set -u
TAG=${1:-r06h}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager --no-alt-precision --no-roofline"
timeout 600 python bench.py $B > $OUT/bench_on.json 2> $OUT/bench_on.err
python - $OUT/bench_on.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')][-1])
print(round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms')
PY
