#!/bin/bash
# GPU run r03k: validation of HEAD (full GPU suite, smoke, driver bench command) + stride-2 dgrad probe, budget, C5 probe.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03k; mkdir -p $OUT
export TMPDIR=/tmp
python tools/s2_dgrad_probe.py > $OUT/s2_dgrad.txt 2> $OUT/s2_dgrad.err
python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python tools/step_budget.py > $OUT/step_budget.txt 2>&1
timeout 600 python tools/c5_probe.py > $OUT/c5_probe.json 2> $OUT/c5_probe.err
head -1 $OUT/s2_dgrad.txt; tail -4 $OUT/pytest.log; tail -2 $OUT/smoke.log; cut -c1-400 $OUT/bench_driver.json; cut -c1-200 $OUT/bench.json; head -8 $OUT/step_budget.txt; cat $OUT/c5_probe.json
