#!/bin/bash
# round-5 closing records, part 1: GPU suite on the final sources + every PMC traffic pass (Winograd roofline launch, direct
# kernels of the same layer with HG_WINO=0, dense histogram kernels, thresholding) + FETCH_SIZE calibration.
set -u
TAG=${1:-r05final2}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
cp gpurun_out/c3_parity.json gpurun_out/c5_parity.json $OUT/ 2>/dev/null
timeout 500 bash tools/wino_pmc.sh gpurun_out/$TAG/pmc > $OUT/pmc.log 2>&1; cat $OUT/pmc/wino_roofline_traffic.txt; grep "k_wino" $OUT/pmc/wino_roofline_kernel_stats.csv | cut -c1-170; grep "k_wino" $OUT/pmc/wino_leading_kernel_stats.csv | cut -c1-170
HG_WINO=0 timeout 400 bash tools/conv_traffic.sh gpurun_out/$TAG/conv > $OUT/conv.log 2>&1; tail -30 $OUT/conv.log | cut -c1-170
HG_HIST_METHOD=thresholding timeout 300 bash tools/hist_traffic.sh gpurun_out/$TAG/thr > $OUT/thr.log 2>&1; tail -16 $OUT/thr.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib/fetch -o p -- $ROOT/tools/ubench/fetch_calib > $OUT/fetch_calib_times.txt 2>&1)
python tools/pmc_summary.py $OUT/calib "" > $OUT/fetch_calib_pmc.txt 2>&1
find "$OUT" -name "*.csv" -size +300k -delete
