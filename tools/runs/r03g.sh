#!/bin/bash
# GPU run r03g: full validation + the records of round 3 (bench, timeline, budget, kernel stats, PMC traffic).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
python tools/step_budget.py > $OUT/step_budget.txt 2>&1
bash tools/conv_traffic.sh gpurun_out/r03g/conv_traffic > $OUT/conv_traffic.log 2>&1
HG_HIST_METHOD=thresholding bash tools/hist_traffic.sh gpurun_out/r03g/thr_traffic > $OUT/thr_traffic.log 2>&1
python bench.py --workload hist --cpu-images 2 --cpu-reps 3 > $OUT/bench_hist.json 2> $OUT/bench_hist.err
python bench.py --workload rehistogan > $OUT/bench_rehistogan.json 2> $OUT/bench_rehistogan.err
tail -4 $OUT/pytest.log; cut -c1-300 $OUT/bench.json; head -6 $OUT/step_timeline.txt; head -8 $OUT/step_budget.txt; cat $OUT/conv_traffic/traffic.txt | head -30; cat $OUT/thr_traffic/traffic.txt | head -20
