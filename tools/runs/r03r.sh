#!/bin/bash
# GPU run r03r: demodulation style-gradient kernel -- unit tests, A/B, trace of the long library launches, network tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03r; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_conv_gpu.py -m gpu -q -x -p no:cacheprovider -k "demod" > $OUT/pytest_unit.log 2>&1; tail -3 $OUT/pytest_unit.log
python tools/sched_probe.py --toggle histogan_amd.ops:FUSED_DEMOD_BWD --rounds 4 > $OUT/ab_demod_style.json 2> $OUT/ab.err; cat $OUT/ab_demod_style.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 12 > $OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/trace_long_aten.py $DB 40 > $OUT/long_aten.txt 2>&1
rm -rf $OUT/trace
cut -c1-200 $OUT/long_aten.txt | tail -25
python -m pytest tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_graph_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
