#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command itself -> gpurun_out/r04l/bench_kernel_stats.txt (copied to profiles/)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04l; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench.json 2> $OUT/bench.err)
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python tools/kstats.py $OUT/trace | head -40 > $OUT/bench_kernel_stats.txt
cat $OUT/bench_kernel_stats.txt | head -14
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04l/bench.json').read().splitlines() if l.strip().startswith('{')][-1])
print('under rocprofv3:', round(d['value'],1), 'images/s; roofline launch', d['roofline']['launch_ms'], 'ms', d['roofline']['frac'])
PY
find $OUT -name "*.csv" -size +300k -delete
