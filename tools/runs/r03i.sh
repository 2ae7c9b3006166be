#!/bin/bash
# GPU run r03i: scheduling A/B (early D step, stream priorities) + the full budget table.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03i; mkdir -p $OUT
export TMPDIR=/tmp
python tools/sched_probe.py --toggle D_STEP_EARLY > $OUT/ab_dstep.json 2> $OUT/ab_dstep.err
for cfg in "" "HG_G_STREAM_PRIO=-1" "HG_W_STREAM_PRIO=-1" "HG_G_STREAM_PRIO=-1 HG_W_STREAM_PRIO=-1" ""; do
  env $cfg python tools/sched_probe.py --rounds 2 >> $OUT/ab_prio.json 2>> $OUT/ab_prio.err
done
HG_BUDGET_ROWS=400 python tools/step_budget.py > $OUT/step_budget_all.txt 2>&1
cat $OUT/ab_dstep.json $OUT/ab_prio.json; tail -3 $OUT/ab_dstep.err
