#!/bin/bash
# GPU run r04d: (1) reference-golden test at C3 width; (2) K-split combine variants (tagged builds xcd1 / xcd2) against the
# default; (3) bf16x6 end-to-end evaluation of the parity suites; (4) histogram workgroup-count sweep; (5) configs[4] at
# full width: test + bench line.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04d; mkdir -p $OUT
export TMPDIR=/tmp
P="python -m pytest -m gpu -q -p no:cacheprovider"
timeout 300 $P -x tests/test_c3_parity_gpu.py::test_c3_networks_match_reference_golden > $OUT/golden.log 2>&1; tail -3 $OUT/golden.log
cp gpurun_out/c3_parity.json $OUT/c3_parity_golden.json 2>/dev/null

echo "== K-split combine variants"
for tag in xcd1 xcd2; do
  HG_LIB_TAG=$tag timeout 200 $P -x tests/test_conv_gpu.py -k "conv" > $OUT/pytest_$tag.log 2>&1; tail -1 $OUT/pytest_$tag.log
  HG_LIB_TAG=$tag timeout 120 python tools/sched_probe.py --rounds 3 > $OUT/step_$tag.json 2> $OUT/step_$tag.err; cat $OUT/step_$tag.json
done
timeout 120 python tools/sched_probe.py --rounds 3 > $OUT/step_default.json 2> $OUT/step_default.err; cat $OUT/step_default.json

echo "== bf16x6 end to end"
rm -f gpurun_out/c3_parity.json
HG_CONV_PRECISION=b6 timeout 600 $P --tb=line tests/test_c3_parity_gpu.py tests/test_nets_gpu.py > $OUT/pytest_b6.log 2>&1; tail -25 $OUT/pytest_b6.log
cp gpurun_out/c3_parity.json $OUT/c3_parity_b6.json 2>/dev/null
HG_CONV_PRECISION=b6 timeout 120 python tools/sched_probe.py --rounds 2 > $OUT/step_b6.json 2> $OUT/step_b6.err; cat $OUT/step_b6.json

echo "== histogram workgroup targets"
for w in 512 768 1024 1536 2048; do
  HG_FWD_WGS=$w HG_BWD_WGS=$w HG_HIST_ITERS=30 timeout 100 python tools/hist_probe.py 2>&1 | tail -1 | sed "s/^/wgs=$w: /"
done | tee $OUT/hist_wgs.txt

echo "== configs[4] at full width"
timeout 600 $P -x tests/test_c5_sanity_gpu.py > $OUT/pytest_c5.log 2>&1; tail -3 $OUT/pytest_c5.log
timeout 600 python bench.py --workload c5 --steps 8 --warmup 2 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 1500 $OUT/bench_c5.json; tail -3 $OUT/bench_c5.err
