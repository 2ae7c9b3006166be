#!/bin/bash
# GPU run r04h: grouped style projections: parity tests, step A/B (module flag toggled between interleaved blocks), whole nets suites
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04h; mkdir -p $OUT
export TMPDIR=/tmp
P="python -m pytest -m gpu -q -p no:cacheprovider"
timeout 300 $P -x tests/test_linear_gpu.py > $OUT/pytest_linear.log 2>&1; tail -5 $OUT/pytest_linear.log
timeout 900 $P -x tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_graph_gpu.py tests/test_ddp_step_gpu.py tests/test_rehistogan_gpu.py > $OUT/pytest_nets.log 2>&1; tail -4 $OUT/pytest_nets.log
timeout 200 python tools/sched_probe.py --rounds 3 --toggle histogan_amd.ops:GROUPED_STYLES > $OUT/ab_grouped.json 2> $OUT/ab_grouped.err; cat $OUT/ab_grouped.json; tail -2 $OUT/ab_grouped.err
