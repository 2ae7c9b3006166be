#!/bin/bash
# GPU run r04g: the whole GPU suite, the b6 run of the plain-step test, the default bench line
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=15 > $OUT/pytest_all.log 2>&1; tail -25 $OUT/pytest_all.log
cp gpurun_out/c3_parity.json $OUT/c3_parity.json 2>/dev/null
HG_CONV_PRECISION=b6 timeout 300 python -m pytest -m gpu -q -p no:cacheprovider "tests/test_c3_parity_gpu.py::test_c3_train_step_matches_oracle" > $OUT/pytest_b6_steps.log 2>&1; tail -2 $OUT/pytest_b6_steps.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench_default.json'))
print(d['value'], d['ms_per_step'], d['schedule_mix'], d.get('alt_precision'), d['roofline']['frac'], d['roofline']['hist']['frac'], d['roofline']['hist']['fwd']['frac'], d['cpu_baseline']['value'], d['reference_eager_rocm'].get('value'))
PY
tail -3 $OUT/bench_default.err
