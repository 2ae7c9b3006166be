#!/usr/bin/env python3
"""A few Winograd launches in isolation (hg_wino_conv2d / hg_wino_wgrad) for rocprofv3 --pmc passes and kernel traces:
   HG_ONE_ITERS launches each of the output convolution at 256->128 @64^2 (the roofline layer), 512->512 @16^2, and the
   weight gradient at 512->512 @16^2 and 128->128 @64^2 (batch 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histogan_amd import conv as C
dev = torch.device('cuda:0')
B = 32
it = int(os.environ.get('HG_ONE_ITERS', 4))
ONLY = os.environ.get('HG_PMC_ONLY', '')      # 'roofline': bench.py's roofline launch alone (per-launch FETCH_SIZE / WRITE_SIZE)
# HG_PMC_WARM_MS: that many ms of library GEMMs first (other kernel names).  A kernel trace that starts on an idle GPU shows a
# power-management transient in its first ~60 launches (367 -> 407 -> 324 us and still falling at 256 -> 128 @64^2; a GEMM
# warm-up only moves the transient: 421 -> 338), not the steady state bench.py times after its training steps: the traces that
# are compared with bench.py run 600 launches instead (tools/wino_pmc.sh)
WARM = float(os.environ.get('HG_PMC_WARM_MS', 0))
if WARM > 0:
    m = torch.randn(4096, 4096, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(20):
            m @ m
        e1.record()
        e1.synchronize()
        if e0.elapsed_time(e1) >= WARM:
            break
FWD = ((256, 128, 64),) if ONLY == 'roofline' else ((256, 128, 64), (512, 512, 16))
WG = ((256, 128, 64),) if ONLY == 'roofline' else ((512, 512, 16), (128, 128, 64))
if ONLY == 'leading':
    # bench.py's `leading_kernels` launches, one kernel-trace line each: the data gradient at 512 -> 256 @32^2 and the
    # no-autograd generator stage (modulate + conv + demodulate + noise + LeakyReLU in one hg_wino_conv2d) at 256 -> 128 @64^2
    from histogan_amd import ops
    K, N, S = 512, 256, 32
    w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
    go = torch.randn(B, N, S, S, device=dev)
    ud = C._wino_pack(w, C.PACK_DGRAD)
    for _ in range(it):
        C.wino_conv(go, ud, K)
    K, N, S = 256, 128, 64
    with torch.no_grad():
        x = torch.randn(B, K, S, S, device=dev)
        st = 0.3 * torch.randn(B, K, device=dev)
        w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
        nzt = torch.rand(B, 256, 256, device=dev)
        wn, bn = torch.randn(N, device=dev), torch.randn(N, device=dev)
        for _ in range(it):
            ops.modconv_stage(x, st, w, nzt, wn, bn, demod=True, upsample=False, act=True)
    torch.cuda.synchronize()
    sys.exit(0)
for K, N, S in FWD:
    x = torch.randn(B, K, S, S, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
    u = C._wino_pack(w, C.PACK_FWD)
    for _ in range(it):
        C.wino_conv(x, u, N)
for K, N, S in WG:
    x = torch.randn(B, K, S, S, device=dev)
    go = torch.randn(B, N, S, S, device=dev)
    assert C.wino_wgrad_supported(B, K, N, S, S)
    for _ in range(it):
        C.conv_wgrad(x, go, 3)
torch.cuda.synchronize()
