#!/bin/bash
# GPU run r03s: logit layer as product + row sum (A/B through SKINNY_SPLIT), network tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03s; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sched_probe.py --toggle histogan_amd.ops:SKINNY_SPLIT --rounds 4 > $OUT/ab_skinny.json 2> $OUT/ab.err; cat $OUT/ab_skinny.json
python -m pytest tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_ddp_step_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
