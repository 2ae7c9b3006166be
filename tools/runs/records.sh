#!/bin/bash
# The records run of a round (round 4: gpurun_out/r04i): full GPU suite, smoke, PMC traffic passes, step budget / timeline /
# kernel stats, the bench lines.  usage (GPU box, repo root):  bash tools/runs/records.sh [tag]      -> gpurun_out/<tag>/
# Afterwards, here: python tools/make_traffic_record.py gpurun_out/<tag>/conv_traffic/traffic.txt gpurun_out/<tag>/thr_traffic/traffic.txt <commit>
# and copy what should be judged into profiles/.  The one-off A/B runs of the round are env knobs over
# tools/sched_probe.py, tools/hist_probe.py, tools/hist_cycles.py, tools/ab_step.py.
set -u
TAG=${1:-records}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash tools/conv_traffic.sh gpurun_out/$TAG/conv_traffic > $OUT/conv_traffic.log 2>&1; tail -12 $OUT/conv_traffic.log
HG_HIST_METHOD=thresholding bash tools/hist_traffic.sh gpurun_out/$TAG/thr_traffic > $OUT/thr_traffic.log 2>&1; tail -5 $OUT/thr_traffic.log
timeout 300 python tools/step_budget.py > $OUT/step_budget.txt 2>&1; head -8 $OUT/step_budget.txt; tail -1 $OUT/step_budget.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
head -7 $OUT/step_timeline.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 600 python bench.py --workload hist > $OUT/bench_hist.json 2> $OUT/bench_hist.err
timeout 600 python bench.py --workload rehistogan --no-cpu-baseline > $OUT/bench_rehistogan.json 2> $OUT/bench_rehistogan.err
HG_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 12 --warmup 4 --no-roofline > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err
for f in bench_driver bench_hist bench_rehistogan bench_n2_gloo; do python - $OUT/$f.json <<'PY'
import json,sys
L=[l for l in open(sys.argv[1]).read().splitlines() if l.strip()]
d=json.loads(L[-1]); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'lines', len(L), round(d['value'],1), round(d['ms_per_step'],3), d.get('n_gpus'), r.get('frac'), r.get('traffic'))
PY
done
