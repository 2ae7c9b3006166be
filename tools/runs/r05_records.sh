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
timeout 120 tools/ubench/hist_fwd_loop > $OUT/ubench_hist_fwd_loop.txt 2>&1; head -2 $OUT/ubench_hist_fwd_loop.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --workload hist > $OUT/bench_hist.json 2> $OUT/bench_hist.err
timeout 600 python bench.py --workload rehistogan --no-cpu-baseline > $OUT/bench_rehistogan.json 2> $OUT/bench_rehistogan.err
timeout 900 python bench.py --workload c5 --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_c5.json 2> $OUT/bench_c5.err
HG_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 12 --warmup 4 --no-roofline > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err
for f in bench_driver bench_default bench_hist bench_rehistogan bench_c5 bench_n2_gloo; do python - $OUT/$f.json <<'PY'
import json,sys
L=[l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')]
if not L: print(sys.argv[1].split('/')[-1], 'NO LINE'); sys.exit(0)
d=json.loads(L[-1]); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'lines', len(L), round(d['value'],1), round(d['ms_per_step'],3), d.get('n_gpus'), r.get('frac'), r.get('traffic'))
PY
done
