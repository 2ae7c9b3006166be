#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
python tools/step_budget.py > $OUT/step_budget.txt 2>&1
tail -4 $OUT/pytest.log; cut -c1-300 $OUT/bench.json; head -40 $OUT/step_timeline.txt | cut -c1-130; head -8 $OUT/step_budget.txt
