#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp
python tools/debug_gp.py > $OUT/debug_gp.txt 2>&1
python -m pytest tests/test_c3_parity_gpu.py tests/test_conv_gpu.py tests/test_hist_gpu.py tests/test_hist_big_gpu.py -q -p no:cacheprovider -k "not conv_launch" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/histtrace -o t -- python $ROOT/bench.py --workload hist --no-cpu-baseline --steps 30 > $OUT/bench_hist.json 2> $OUT/bench_hist.err)
find $OUT/histtrace -name "*kernel_stats.csv" | head -1 | xargs -r head -12 > $OUT/hist_kernel_stats.csv
rm -rf $OUT/histtrace
python tools/ab_step.py --rounds 1 > $OUT/ab_step.json 2> $OUT/ab_step.err
cat $OUT/debug_gp.txt | tail -30; tail -5 $OUT/pytest.log; cat $OUT/hist_kernel_stats.csv; cat $OUT/ab_step.json
