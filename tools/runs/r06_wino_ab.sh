#!/bin/bash
# A/B of a k_wino build flag: default library vs libhistogan_hip_<tag>.so (HG_LIB_TAG), same box: Winograd tests, the bench's
# roofline launch and leading kernels, the driver's train line.   usage: bash tools/runs/r06_wino_ab.sh <out tag> <lib tag>
set -u
TAG=${1:-r06wino}; LT=${2:-nobp}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_wino_gpu.py tests/test_conv_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager"
for rep in 1 2; do
timeout 600 python bench.py $B > $OUT/bench_new_$rep.json 2> /dev/null
HG_LIB_TAG=$LT timeout 600 python bench.py $B > $OUT/bench_old_$rep.json 2> /dev/null
done
for f in bench_new_1 bench_old_1 bench_new_2 bench_old_2; do python - $OUT/$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')][-1]); r=d['roofline']
lk=r.get('leading_kernels',[])
print(sys.argv[1].split('/')[-1], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms; roofline launch', round(r['launch_ms']*1e3,1), 'us', round(r['frac'],4), '; dgrad', round(r['dgrad']['launch_ms']*1e3,1) if 'dgrad' in r else None, '; leading', [(k.get('launch_ms') and round(k['launch_ms']*1e3,1), round(k.get('frac',0),3)) for k in lk][:6])
PY
done
