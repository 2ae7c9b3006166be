#!/bin/bash
# GPU run r03x: asynchronous pre-pack of the generator's operands (A/B) + step / graph / checkpoint tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03x; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sched_probe.py --toggle histogan_amd.conv:PREPACK --rounds 4 > $OUT/ab_prepack.json 2> $OUT/ab.err; cat $OUT/ab_prepack.json
python -m pytest tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_graph_gpu.py tests/test_trainer_io_gpu.py tests/test_ddp_step_gpu.py -m gpu -q -x -p no:cacheprovider -k "step or graph or train or load or save or nan" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
