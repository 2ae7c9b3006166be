#!/bin/bash
# round-5 closing records, part 2 (after profiles/r05_pmc_traffic.json exists): the bench lines + the kernel trace of the bench command
set -u
TAG=${1:-r05final3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --workload hist > $OUT/bench_hist.json 2> $OUT/bench_hist.err
timeout 400 bash tools/runs/bench_trace.sh $TAG/trace > $OUT/bench_trace.log 2>&1; head -14 $OUT/trace/bench_kernel_stats.txt 2>/dev/null | cut -c1-160
for f in bench_driver bench_default bench_hist; do python - $OUT/$f.json <<'PY'
import json,sys
L=[l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')]
if not L: print(sys.argv[1].split('/')[-1], 'NO LINE'); sys.exit(0)
d=json.loads(L[-1]); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'lines', len(L), round(d['value'],1), round(d['ms_per_step'],3), d.get('n_gpus'), r.get('frac'), r.get('traffic'), (r.get('wgrad') or {}).get('traffic'))
PY
done
