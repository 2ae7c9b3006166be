#!/bin/bash
# GPU run r04k: execution-mode A/B at HEAD (eager blocking / eager deferred / hipGraph replay) now that a plain step is 1 126
# launches; HG_DNL_KEEP_CONV=0 (one stream less in k_dnl_bwd); default bench line with the re-measured traffic record.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04k; mkdir -p $OUT
timeout 300 python tools/ab_step.py --rounds 2 > $OUT/ab_step.json 2> $OUT/ab_step.err; cat $OUT/ab_step.json
timeout 200 python tools/sched_probe.py --rounds 2 > $OUT/keep1.json 2>/dev/null; cat $OUT/keep1.json
HG_DNL_KEEP_CONV=0 timeout 200 python tools/sched_probe.py --rounds 2 > $OUT/keep0.json 2>/dev/null; cat $OUT/keep0.json
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04k/bench_default.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['hist']['traffic'], r['hist']['thresholding']['traffic'], r.get('leading_kernels'), d['host']['launches_per_plain_step_eager'], d['alt_precision']['images_per_s'])
PY
