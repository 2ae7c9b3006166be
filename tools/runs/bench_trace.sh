#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command itself -> gpurun_out/<tag>/bench_kernel_stats.txt (copied to profiles/)
# usage: bash tools/runs/bench_trace.sh <tag>
set -u
TAG=${1:-bench_trace}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench.json 2> $OUT/bench.err)
python tools/kstats.py $OUT/trace | head -40 > $OUT/bench_kernel_stats.txt
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')][-1])
print('under rocprofv3:', round(d['value'],1), 'images/s; roofline launch', d['roofline']['launch_ms'], 'ms', d['roofline']['frac'])
PY
find $OUT -name "*.csv" -size +300k -delete
