#!/bin/bash
# GPU run r03n: demodulation weight term written straight into the flat gradient slot -- A/B + the affected tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03n; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sched_probe.py --toggle histogan_amd.conv:DIRECT_DEMOD > $OUT/ab_demod.json 2> $OUT/ab_demod.err
python -m pytest tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_graph_gpu.py tests/test_ddp_step_gpu.py tests/test_trainer_io_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1
cat $OUT/ab_demod.json; tail -3 $OUT/ab_demod.err; tail -5 $OUT/pytest.log
