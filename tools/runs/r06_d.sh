#!/bin/bash
set -u
TAG=${1:-r06d}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gstage_gpu.py tests/test_c3_parity_gpu.py -x -q -p no:cacheprovider -k "gstage or generator or train_step" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python tools/phase_probe.py --index 5 > $OUT/phase_plain.json 2> $OUT/phase_plain.err
HG_H_SIDE_GRAD=0 timeout 300 python tools/phase_probe.py --index 5 > $OUT/phase_plain_hside0.json 2> /dev/null
for f in phase_plain phase_plain_hside0; do python - $OUT/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], ' '.join('%s=%.2f'%(p['phase'][:22],p['gpu_ms']) for p in d['phases']), 'step', d['step_gpu_ms_start_to_start'])
PY
done
