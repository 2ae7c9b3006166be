#!/bin/bash
# GPU run r03a: full GPU test suite (new C3 parity / f-2 / f-4 / bench tests), default bench, lazy-stats A/B, step timeline.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=30 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
for lazy in 1 0; do
  HG_LAZY_STATS=$lazy python bench.py --steps 32 --warmup 4 --no-roofline > $OUT/bench_lazy$lazy.json 2> $OUT/bench_lazy$lazy.err
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_timeline.py $DB 12 > $OUT/step_timeline.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
rm -rf $OUT/trace
tail -5 $OUT/pytest.log; cat $OUT/bench_lazy1.json $OUT/bench_lazy0.json | cut -c1-400; head -8 $OUT/step_timeline.txt
