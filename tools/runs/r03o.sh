#!/bin/bash
# GPU run r03o: demodulation weight-term kernel + chunked skinny GEMMs -- tests and A/B.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03o; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_conv_gpu.py -m gpu -q -x -p no:cacheprovider -k "demod_weight_term" > $OUT/pytest_unit.log 2>&1; tail -3 $OUT/pytest_unit.log
python tools/sched_probe.py --toggle histogan_amd.conv:DIRECT_DEMOD > $OUT/ab_demod.json 2> $OUT/ab_demod.err
python tools/sched_probe.py --toggle histogan_amd.ops:SKINNY_SPLIT > $OUT/ab_skinny.json 2> $OUT/ab_skinny.err
python -m pytest tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_graph_gpu.py tests/test_ddp_step_gpu.py tests/test_trainer_io_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1
cat $OUT/ab_demod.json $OUT/ab_skinny.json; tail -3 $OUT/ab_skinny.err; tail -5 $OUT/pytest.log
