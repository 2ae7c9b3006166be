#!/bin/bash
# GPU run r03t: both latent batches through S at once (A/B), tests that replay the step against the oracle.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03t; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sched_probe.py --toggle BATCH_S --rounds 4 > $OUT/ab_batch_s.json 2> $OUT/ab.err; cat $OUT/ab_batch_s.json
python -m pytest tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_graph_gpu.py tests/test_ddp_step_gpu.py -m gpu -q -x -p no:cacheprovider -k "step or graph or train" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
