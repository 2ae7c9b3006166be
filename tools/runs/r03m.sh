#!/bin/bash
# GPU run r03m: C5 probe (1024^2) eager vs graph after the auto-threshold change.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/c5_probe.py > $OUT/c5_probe.json 2> $OUT/c5_probe.err
HG_GRAPH=1 timeout 600 python tools/c5_probe.py > $OUT/c5_probe_graph.json 2> $OUT/c5_probe_graph.err
HG_GRAPH=0 HG_LAZY_STATS=0 timeout 600 python tools/c5_probe.py > $OUT/c5_probe_sync.json 2> $OUT/c5_probe_sync.err
cat $OUT/c5_probe.json $OUT/c5_probe_graph.json $OUT/c5_probe_sync.json
