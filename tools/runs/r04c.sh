#!/bin/bash
# GPU run r04c: shader-cycle breakdown of the dense histogram kernels (probe build)
set -u
OUT=gpurun_out/r04c; mkdir -p $OUT
HG_LIB_TAG=probe timeout 120 python tools/hist_cycles.py 2>&1 | tee $OUT/hist_cycles_share1.json
HG_LIB_TAG=probe HG_FWD_SHARE_RCP=0 timeout 120 python tools/hist_cycles.py 2>&1 | tee $OUT/hist_cycles_share0.json
