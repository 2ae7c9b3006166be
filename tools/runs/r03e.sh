#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03e; mkdir -p $OUT
python tools/debug_gp4.py > $OUT/debug_gp4.txt 2>&1
tail -22 $OUT/debug_gp4.txt
