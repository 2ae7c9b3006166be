#!/bin/bash
set -u
TAG=${1:-r06f}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gstage_gpu.py tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_trainer_io_gpu.py tests/test_graph_gpu.py tests/test_f2_f4_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], ' '.join('%s=%.2f'%(p['phase'][:24],p['gpu_ms']) for p in d['phases']), 'step', d['step_gpu_ms_start_to_start'])
PY
}
timeout 300 python tools/phase_probe.py --index 5 > $OUT/phase_plain.json 2> $OUT/phase_plain.err; show $OUT/phase_plain.json
timeout 300 python tools/phase_probe.py --index 4 > $OUT/phase_gp.json 2> /dev/null; show $OUT/phase_gp.json
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager --no-alt-precision --no-roofline"
timeout 600 python bench.py $B > $OUT/bench_on.json 2> $OUT/bench_on.err
HG_EARLY_GOPT=1 timeout 600 python bench.py $B > $OUT/bench_early0.json 2> /dev/null
HG_GFUSED=0 HG_EARLY_GOPT=0 HG_H_SIDE_GRAD=0 timeout 600 python bench.py $B > $OUT/bench_off.json 2> /dev/null
for f in bench_on bench_early0 bench_off; do python - $OUT/$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')][-1])
print(sys.argv[1].split('/')[-1], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms', d.get('schedule_mix',{}).get('ms_plain'), d.get('schedule_mix',{}).get('ms_gp'))
PY
done
