#!/bin/bash
# GPU run r04b: k_hist_fwd with shared reciprocals + asm operand block: parity tests, A/B against the round-3 loop, kernel trace.
set -u
ROOT=$(pwd); OUT=gpurun_out/r04b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_hist_gpu.py tests/test_hist_big_gpu.py tests/test_hist_planes_gpu.py \
  tests/test_c3_parity_gpu.py::test_c3_networks_match_reference_golden > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
cp gpurun_out/c3_parity.json $OUT/c3_parity.json 2>/dev/null
for i in 1 2; do
  HG_FWD_SHARE_RCP=0 HG_HIST_ITERS=40 timeout 100 python tools/hist_probe.py 2>&1 | tail -1 | sed 's/^/share=0: /'
  HG_FWD_SHARE_RCP=1 HG_HIST_ITERS=40 timeout 100 python tools/hist_probe.py 2>&1 | tail -1 | sed 's/^/share=1: /'
done | tee $OUT/ab.txt
HG_HIST_INSZ=150 HG_HIST_ITERS=40 timeout 100 python tools/hist_probe.py 2>&1 | tail -1 | tee -a $OUT/ab.txt
(cd /tmp && HG_HIST_ITERS=40 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -o t -- python "$ROOT/tools/hist_probe.py" > "$ROOT/$OUT/trace.log" 2>&1)
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -r head -8
find "$OUT" -name "*.csv" -size +300k -delete
