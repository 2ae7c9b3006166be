#!/bin/bash
# GPU run r03p: skinny split incl. the histogram vectorizer's first layer (A/B), [fake; real] by two copies; full suite.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03p; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sched_probe.py --toggle histogan_amd.ops:SKINNY_SPLIT --rounds 4 > $OUT/ab_skinny.json 2> $OUT/ab_skinny.err
cat $OUT/ab_skinny.json
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
