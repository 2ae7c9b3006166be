#!/bin/bash
# First GPU run of the next round (prepared at the end of round 3, never executed): the fence-free in-kernel K-split combine
# with a tile's splits adjacent in dispatch order.  Build the tagged libraries HERE first (no GPU needed, ~2 min each):
#   HG_LIB_TAG=xcd1 HG_CFLAGS='-DHG_CONV_XCD_SPLITK=1' python -m histogan_amd.build
#   HG_LIB_TAG=xcd2 HG_CFLAGS='-DHG_CONV_XCD_SPLITK=2' python -m histogan_amd.build
# then:  gpurun --timeout 600 -- 'bash tools/runs/next_xcd_splitk.sh'
# Round-3 numbers to compare with (plain step, C3): two launches 46.6 ms; =1: 48.5 ms (correct; finishers at the tail).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/next_xcd; mkdir -p $OUT
export TMPDIR=/tmp
for tag in xcd1 xcd2; do
  HG_LIB_TAG=$tag timeout 120 python -m pytest tests/test_conv_gpu.py tests/test_c3_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv" > $OUT/pytest_$tag.log 2>&1
  tail -2 $OUT/pytest_$tag.log
  HG_LIB_TAG=$tag timeout 90 python tools/sched_probe.py --rounds 3 > $OUT/step_$tag.json 2> $OUT/step_$tag.err; cat $OUT/step_$tag.json
done
timeout 90 python tools/sched_probe.py --rounds 3 > $OUT/step_default.json 2> $OUT/step_default.err; cat $OUT/step_default.json
timeout 60 tools/ubench/xcd_handoff 1024 4 > $OUT/ubench.txt 2>&1; tail -10 $OUT/ubench.txt
