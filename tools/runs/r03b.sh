#!/bin/bash
# GPU run r03b: tests after the fusions, same-process A/B of the step modes, conv budget, timeline, default bench.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python tools/ab_step.py > $OUT/ab_step.json 2> $OUT/ab_step.err
python tools/step_budget.py > $OUT/step_budget.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/pytest.log; cat $OUT/ab_step.json; head -8 $OUT/step_budget.txt; head -6 $OUT/step_timeline.txt; cut -c1-300 $OUT/bench.json
