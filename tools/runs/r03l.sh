#!/bin/bash
# GPU run r03l: the driver's bench command (fresh processes), graph auto-decision, conv PMC traffic for the record.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03l; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver1.json 2> $OUT/bench_driver1.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager > $OUT/bench_driver2.json 2> $OUT/bench_driver2.err
python -m pytest tests/test_graph_gpu.py tests/test_bench_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1
bash tools/conv_traffic.sh gpurun_out/r03l/conv_traffic > $OUT/conv_traffic.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
for f in bench_driver1 bench_driver2 bench; do python - $OUT/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['value'],1), round(d['ms_per_step'],2), d['schedule_mix'], d['host']['graph_replayed_steps'])
PY
done
tail -3 $OUT/pytest.log; head -20 $OUT/conv_traffic/traffic.txt
