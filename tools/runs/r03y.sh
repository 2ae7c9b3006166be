#!/bin/bash
# GPU run r03y: final validation of HEAD -- full GPU suite, smoke, the driver's bench command, default bench.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03y; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python bench.py --no-reference-eager > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/pytest.log; tail -1 $OUT/smoke.log
for f in bench_driver bench; do python - $OUT/$f.json <<'PY'
import json,sys
L=[l for l in open(sys.argv[1]).read().splitlines() if l.strip()]
d=json.loads(L[-1]); print(sys.argv[1].split('/')[-1], 'lines', len(L), round(d['value'],1), round(d['ms_per_step'],2), d['schedule_mix'], d['host']['graph_replayed_steps'], round(d['roofline']['frac'],3), d['roofline']['traffic'])
PY
done
