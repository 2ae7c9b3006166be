#!/bin/bash
# GPU run r03v: validation of HEAD -- full GPU suite, smoke, the driver's bench command, default bench, budget, timeline.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03v; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
tail -4 $OUT/pytest.log; tail -2 $OUT/smoke.log
for f in bench_driver bench; do python - $OUT/$f.json <<'PY'
import json,sys
L=[l for l in open(sys.argv[1]).read().splitlines() if l.strip()]
d=json.loads(L[-1]); print(sys.argv[1].split('/')[-1], 'lines', len(L), round(d['value'],1), round(d['ms_per_step'],2), d['schedule_mix'], d['host']['graph_replayed_steps'], d['roofline']['frac'], d['roofline']['traffic'])
PY
done
head -6 $OUT/step_timeline.txt
