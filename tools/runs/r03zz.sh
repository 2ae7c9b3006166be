#!/bin/bash
# GPU run r03zz: kernel stats of the train step on the tagged fence-free K-split build (to compare with profiles/r03_train_c3_kernel_stats.md)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03zz; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && HG_LIB_TAG=xcd timeout 100 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats_xcd.md > /dev/null 2>&1
rm -rf $OUT/trace
head -30 $OUT/kernel_stats_xcd.md | cut -c1-150
