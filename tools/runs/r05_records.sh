#!/bin/bash
# round-5 records: full GPU suite, step budget / timeline / kernel stats, Winograd PMC passes.  usage: bash tools/runs/r05_records.sh [tag]
set -u
TAG=${1:-r05rec}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
fi
timeout 300 python tools/step_budget.py > $OUT/step_budget.txt 2>&1; head -8 $OUT/step_budget.txt; tail -1 $OUT/step_budget.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
head -45 $OUT/step_timeline.txt
if [ "${SKIP_PMC:-0}" != "1" ]; then
timeout 500 bash tools/wino_pmc.sh gpurun_out/$TAG/pmc > $OUT/pmc.log 2>&1; grep -A3 "FETCH_SIZE\|WRITE_SIZE" $OUT/pmc/wino_pmc.txt | head -30
fi
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 400 $OUT/bench_driver.json
