#!/bin/bash
# A/B of the XCD-aware work-item order of k_wino / k_wino_wgrad (HG_WINO_XCD=0/1): per-layer probe, step bench, parity tests,
# FETCH / WRITE of the roofline launch.  usage: bash tools/runs/r05_xcd_ab.sh <tag>
set -u
TAG=${1:-r05xcd}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_wino_gpu.py -q -p no:cacheprovider > $OUT/pytest_wino.log 2>&1; tail -2 $OUT/pytest_wino.log
for x in 0 1; do
  HG_WINO_XCD=$x timeout 240 python tools/wino_probe.py --tag xcd$x > $OUT/probe_xcd${x}.txt 2>&1; tail -1 $OUT/probe_xcd${x}.txt
done
for x in 0 1 0 1; do
  HG_WINO_XCD=$x timeout 300 python bench.py --steps 32 --warmup 6 --no-roofline --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_xcd${x}_$RANDOM.json 2>/dev/null
done
for f in $OUT/bench_xcd*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f'.split('/')[-1], round(d['value'],1), round(d['schedule_mix']['ms_plain'],2))"; done
runr() { local x=$1 name=$2; shift 2; (cd /tmp && HG_WINO_XCD=$x HG_PMC_ONLY=roofline timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/roof$x/$name" -o p -- python "$ROOT/tools/wino_pmc.py" > "$OUT/roof${x}_$name.log" 2>&1); }
for x in 0 1; do
  runr $x fetch FETCH_SIZE
  runr $x write WRITE_SIZE
  python tools/pmc_summary.py "$OUT/roof$x" "k_wino" > "$OUT/wino_roofline_traffic_xcd$x.txt" 2>&1
  (cd /tmp && HG_WINO_XCD=$x HG_PMC_ONLY=roofline HG_ONE_ITERS=60 HG_PMC_WARM_MS=1500 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/roof$x/trace" -o t -- python "$ROOT/tools/wino_pmc.py" > "$OUT/roof${x}_trace.log" 2>&1)
  grep "k_wino" $OUT/roof$x/trace/*kernel_stats.csv | cut -c1-200
  grep -A3 "^## " "$OUT/wino_roofline_traffic_xcd$x.txt" | grep -v "^--"
done
find "$OUT" -name "*.csv" -size +300k -delete
