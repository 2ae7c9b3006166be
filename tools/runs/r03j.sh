#!/bin/bash
# GPU run r03j: stride-2 data gradient, merged parity launch on all maps + XCD-paired block order (A/B) and its tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03j; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "HG_DGRAD_S2_MERGE=1 HG_PARITY_XCD=0" "HG_DGRAD_S2_MERGE=1" "HG_DGRAD_S2_MERGE=2 HG_PARITY_XCD=0" "HG_DGRAD_S2_MERGE=2"; do
  env $cfg python tools/s2_dgrad_probe.py >> $OUT/s2_dgrad.txt 2>> $OUT/s2_dgrad.err
done
python -m pytest tests/test_conv_gpu.py tests/test_c3_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv" > $OUT/pytest.log 2>&1
cat $OUT/s2_dgrad.txt; tail -3 $OUT/pytest.log
