#!/bin/bash
# round 6, first look: kernel SEQUENCE of one plain step (with the host's launch calls), eager vs hipGraph A/B incl. single-stream
# capture, the driver's bench command as this box's baseline.   -> gpurun_out/r06a/
set -u
TAG=${1:-r06a}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 5 --steps 8 > $OUT/trace.log 2>&1)
python tools/step_sequence.py $OUT/trace 2 > $OUT/step_sequence_plain.txt 2> $OUT/step_sequence.err
tail -2 $OUT/step_sequence_plain.txt
rm -rf $OUT/trace
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/step_kernels.py --index 4 --steps 6 > $OUT/trace_gp.log 2>&1)
python tools/step_sequence.py $OUT/trace 2 > $OUT/step_sequence_gp.txt 2>> $OUT/step_sequence.err
tail -2 $OUT/step_sequence_gp.txt
rm -rf $OUT/trace
timeout 400 python tools/ab_step.py --steps 12 --rounds 2 > $OUT/ab_step.json 2> $OUT/ab_step.err; tail -1 $OUT/ab_step.json
HG_G_OVERLAP=0 HG_WGRAD_STREAM=0 HG_H_SIDE=0 timeout 400 python tools/ab_step.py --steps 12 --rounds 2 > $OUT/ab_step_1stream.json 2> $OUT/ab_step_1stream.err; tail -1 $OUT/ab_step_1stream.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-eager --no-alt-precision > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 600 $OUT/bench_driver.json
