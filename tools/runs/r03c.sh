#!/bin/bash
# GPU run r03c: after fixing the in-kernel K-split combine; hist projection pre-pass; thresholding single launch.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03c; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python tools/step_budget.py > $OUT/step_budget.txt 2>&1
HG_CONV_SPLITK_INKERNEL=0 python tools/step_budget.py > $OUT/step_budget_two_launch.txt 2>&1
python tools/ab_step.py > $OUT/ab_step.json 2> $OUT/ab_step.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload hist --cpu-images 2 --cpu-reps 3 > $OUT/bench_hist.json 2> $OUT/bench_hist.err
HG_THR_ONE_LAUNCH=0 python bench.py --workload hist --no-cpu-baseline > $OUT/bench_hist_thr2.json 2> $OUT/bench_hist_thr2.err
tail -4 $OUT/pytest.log; cat $OUT/ab_step.json; head -8 $OUT/step_budget.txt; head -8 $OUT/step_budget_two_launch.txt; cut -c1-300 $OUT/bench.json
