#!/bin/bash
# GPU run r03q: kernel trace of a few steps -> long library launches of one plain step, timeline, kernel stats.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03q; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_long_aten.py $DB 12 > $OUT/long_aten.txt 2>&1
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
cat $OUT/long_aten.txt | cut -c1-230; head -6 $OUT/step_timeline.txt
