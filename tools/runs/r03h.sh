#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03h; mkdir -p $OUT
python -m pytest tests/test_hist_gpu.py tests/test_hist_big_gpu.py tests/test_hist_planes_gpu.py -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
for i in 1 2; do
python bench.py --workload hist --no-cpu-baseline --steps 40 > $OUT/bench_hist_kg1_$i.json 2> $OUT/bench_hist_kg1_$i.err
HG_LIB_TAG=kg0 python bench.py --workload hist --no-cpu-baseline --steps 40 > $OUT/bench_hist_kg0_$i.json 2> $OUT/bench_hist_kg0_$i.err
done
tail -3 $OUT/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03h/bench_hist_kg*.json')):
    s=open(f).read(); d=json.loads(s[s.index('{"metric'):]); r=d['roofline']
    print(f.split('/')[-1], 'bwd ms', round(r['launch_ms'],4), 'frac', round(r['frac'],4), 'fwd ms', round(r['fwd']['launch_ms'],4), round(r['fwd']['frac'],4), 'value', round(d['value']))
PY
