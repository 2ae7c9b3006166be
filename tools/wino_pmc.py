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
for K, N, S in ((256, 128, 64), (512, 512, 16)):
    x = torch.randn(B, K, S, S, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
    u = C._wino_pack(w, C.PACK_FWD)
    for _ in range(it):
        C.wino_conv(x, u, N)
for K, N, S in ((512, 512, 16), (128, 128, 64)):
    x = torch.randn(B, K, S, S, device=dev)
    go = torch.randn(B, N, S, S, device=dev)
    assert C.wino_wgrad_supported(B, K, N, S, S)
    for _ in range(it):
        C.conv_wgrad(x, go, 3)
torch.cuda.synchronize()
