#!/bin/bash
# round-6 records: step budget, phase probe (un-profiled), traced timeline / kernel stats, PMC traffic passes, bench lines.
# usage (GPU box, repo root): bash tools/runs/r06_records.sh [tag]     -> gpurun_out/<tag>/ ; then, in the authoring container:
#   python tools/make_traffic_record_rounds.py gpurun_out/<tag>/pmc/wino_roofline_traffic.txt gpurun_out/<tag>/conv_traffic/traffic.txt gpurun_out/<tag>/thr_traffic/traffic.txt <commit> 06
set -u
TAG=${1:-r06rec}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-1}" != "1" ]; then
timeout 1500 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
fi
timeout 300 python tools/step_budget.py > $OUT/step_budget.txt 2>&1; head -8 $OUT/step_budget.txt; tail -1 $OUT/step_budget.txt
for i in 5 4 0; do timeout 300 python tools/phase_probe.py --index $i > $OUT/phase_$i.json 2> $OUT/phase_$i.err; done
python - $OUT <<'PY' > $OUT/phase_probe.txt
import json,sys
print('# tools/phase_probe.py: phases of the C3 train step on un-profiled HIP events (GPU ms, median of 12 steps), host enqueue ms beside it')
for idx,name in ((5,'plain step'),(4,'gradient-penalty step'),(0,'GP + path-length step')):
    try: d=json.loads(open(f'{sys.argv[1]}/phase_{idx}.json').read().strip().splitlines()[-1])
    except Exception as e: print(name,'missing',e); continue
    print(f'\n== {name} (step index {idx}): {d["step_gpu_ms_start_to_start"]} ms start to start, host lead at the end {d["host_lead_ms_at_end"]} ms')
    for p in d['phases']: print(f'  {p["phase"]:34s} gpu {p["gpu_ms"]:8.3f}   host {p["host_ms"]:8.3f}')
PY
cat $OUT/phase_probe.txt | head -20
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
head -40 $OUT/step_timeline.txt
if [ "${SKIP_PMC:-0}" != "1" ]; then
timeout 600 bash tools/wino_pmc.sh gpurun_out/$TAG/pmc > $OUT/pmc.log 2>&1; grep -A3 "FETCH_SIZE\|WRITE_SIZE" $OUT/pmc/wino_roofline_traffic.txt | head -20
bash tools/conv_traffic.sh gpurun_out/$TAG/conv_traffic > $OUT/conv_traffic.log 2>&1; tail -12 $OUT/conv_traffic.log
HG_HIST_METHOD=thresholding bash tools/hist_traffic.sh gpurun_out/$TAG/thr_traffic > $OUT/thr_traffic.log 2>&1; tail -5 $OUT/thr_traffic.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --workload hist > $OUT/bench_hist.json 2> $OUT/bench_hist.err
timeout 600 python bench.py --workload rehistogan --no-cpu-baseline > $OUT/bench_rehistogan.json 2> $OUT/bench_rehistogan.err
timeout 900 python bench.py --workload c5 --no-cpu-baseline --no-reference-eager > $OUT/bench_c5.json 2> $OUT/bench_c5.err
HG_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 12 --warmup 4 --no-roofline > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err
bash tools/runs/bench_trace.sh $TAG/bench_trace > $OUT/bench_trace.log 2>&1; tail -2 $OUT/bench_trace.log
for f in bench_driver bench_default bench_hist bench_rehistogan bench_c5 bench_n2_gloo; do python - $OUT/$f.json <<'PY'
import json,sys
L=[l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')]
if not L: print(sys.argv[1].split('/')[-1], 'NO LINE'); sys.exit(0)
d=json.loads(L[-1]); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'lines', len(L), round(d['value'],1), round(d['ms_per_step'],3), d.get('n_gpus'), r.get('frac'), r.get('traffic'))
PY
done
