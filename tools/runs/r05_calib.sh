#!/bin/bash
# FETCH_SIZE calibration (tools/ubench/fetch_calib) + the step bench A/B of HG_WINO_XCD.  usage: bash tools/runs/r05_calib.sh <tag>
set -u
TAG=${1:-r05calib}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 tools/ubench/fetch_calib > $OUT/fetch_calib_times.txt 2>&1; cat $OUT/fetch_calib_times.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib/fetch -o p -- $ROOT/tools/ubench/fetch_calib > $OUT/calib.log 2>&1)
python tools/pmc_summary.py $OUT/calib "" > $OUT/fetch_calib_pmc.txt 2>&1; cat $OUT/fetch_calib_pmc.txt
find "$OUT" -name "*.csv" -size +300k -delete
