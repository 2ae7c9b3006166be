#!/bin/bash
# GPU run r03w: coalesced gq fill of k_demod_style_grad -- unit tests, isolated timing, step A/B is not needed (kernel was hidden).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03w; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_conv_gpu.py -m gpu -q -x -p no:cacheprovider -k "demod" 2>&1 | tail -2
python - <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from histogan_amd._lib import lib, check, raw_stream
dev = torch.device('cuda:0')
for (B, N, K) in ((32, 2048, 2048), (32, 1024, 2048), (32, 512, 1024), (32, 32, 64)):
    wsq = torch.rand(N, K, device=dev); s1 = torch.rand(B, K, device=dev); gd = torch.randn(B, N, device=dev); d = torch.rand(B, N, device=dev)
    gy = torch.empty(B, K, device=dev); nb = lib.hg_demod_style_grad_workspace_bytes(B, N, K); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    f = lambda: check(lib.hg_demod_style_grad(gd.data_ptr(), d.data_ptr(), s1.data_ptr(), wsq.data_ptr(), gy.data_ptr(), B, N, K, ws.data_ptr(), nb, raw_stream(dev)), 'x')
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(B, N, K, '%.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
PY
python tools/sched_probe.py --rounds 2 2>/dev/null
