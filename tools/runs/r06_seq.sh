#!/bin/bash
# kernel-trace-only sequence of one plain step (no HIP API trace: the host is not slowed down)  -> gpurun_out/<tag>/
set -u
TAG=${1:-r06seq}; IDX=${2:-5}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index $IDX --steps 8 > $OUT/trace.log 2>&1)
python tools/step_sequence.py $OUT/trace 2 > $OUT/step_sequence_$IDX.txt 2> $OUT/step_sequence.err
tail -2 $OUT/step_sequence_$IDX.txt
rm -rf $OUT/trace
