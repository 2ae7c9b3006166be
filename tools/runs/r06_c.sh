#!/bin/bash
# fused generator backward: tests, phase probe, bench A/B   -> gpurun_out/<tag>/
set -u
TAG=${1:-r06c}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gstage_gpu.py tests/test_nets_gpu.py -x -q -p no:cacheprovider > $OUT/pytest_a.log 2>&1; tail -5 $OUT/pytest_a.log
timeout 300 python tools/phase_probe.py --index 5 > $OUT/phase_plain.json 2> $OUT/phase_plain.err; tail -c 1500 $OUT/phase_plain.json
HG_GFUSED=0 timeout 300 python tools/phase_probe.py --index 5 > $OUT/phase_plain_off.json 2> /dev/null; tail -c 1500 $OUT/phase_plain_off.json
timeout 900 python -m pytest tests/test_c3_parity_gpu.py -x -q -p no:cacheprovider > $OUT/pytest_c3.log 2>&1; tail -5 $OUT/pytest_c3.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager --no-alt-precision --no-roofline > $OUT/bench_on.json 2> $OUT/bench_on.err; tail -c 400 $OUT/bench_on.json
HG_GFUSED=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager --no-alt-precision --no-roofline > $OUT/bench_off.json 2> $OUT/bench_off.err; tail -c 400 $OUT/bench_off.json
