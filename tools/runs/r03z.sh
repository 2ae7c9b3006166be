#!/bin/bash
# GPU run r03z (last GPU seconds of round 3): the fence-free in-kernel K-split combine (tagged build, -DHG_CONV_XCD_SPLITK=1).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03z; mkdir -p $OUT
export TMPDIR=/tmp
HG_LIB_TAG=xcd timeout 50 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -p no:cacheprovider -k "splitk or 2048 or stride2" > $OUT/pytest_xcd.log 2>&1; tail -3 $OUT/pytest_xcd.log
HG_LIB_TAG=xcd timeout 60 python tools/sched_probe.py --rounds 2 > $OUT/step_xcd.json 2> $OUT/step_xcd.err; cat $OUT/step_xcd.json
timeout 60 python tools/sched_probe.py --rounds 2 > $OUT/step_default.json 2> $OUT/step_default.err; cat $OUT/step_default.json
