#!/bin/bash
# GPU run r03u: style projections ahead on a second stream, H beside S (A/B each), generator / step tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03u; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sched_probe.py --toggle histogan_amd.nets:STYLES_AHEAD --rounds 4 > $OUT/ab_styles_ahead.json 2> $OUT/ab.err; cat $OUT/ab_styles_ahead.json
python tools/sched_probe.py --toggle H_SIDE --rounds 4 > $OUT/ab_h_side.json 2>> $OUT/ab.err; cat $OUT/ab_h_side.json
python -m pytest tests/test_nets_gpu.py tests/test_c3_parity_gpu.py tests/test_graph_gpu.py tests/test_trainer_io_gpu.py tests/test_f2_f4_gpu.py -m gpu -q -x -p no:cacheprovider -k "step or graph or train or generat or evaluat" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
