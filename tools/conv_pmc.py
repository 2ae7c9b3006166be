#!/usr/bin/env python3
"""Runs one forward / dgrad / wgrad launch per tile shape of the conv kernels (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histogan_amd import conv as C
dev = torch.device('cuda:0')
B = 32
# (K, N, S, k): 128x128 tile, 64x64 tile, 64x256 tile, 32x256 tile, 64x64 split-K
for K, N, S, k in [(256, 128, 64, 3), (512, 512, 16, 3), (128, 64, 128, 3), (64, 32, 256, 3), (2048, 2048, 2, 3)]:
    x = torch.randn(B, K, S, S, device=dev)
    w = torch.randn(N, K, k, k, device=dev) / (K * k * k) ** 0.5
    go = torch.randn(B, N, S, S, device=dev)
    wf = C.pack_weights(w, C.PACK_FWD)
    for _ in range(3):
        C.conv_fwd_packed(x, wf, N, k)
        C.conv_wgrad(x, go, k)
torch.cuda.synchronize()
