#!/usr/bin/env python3
"""k_conv throughput against the number of blocks per launch (batch sweep) -- run under HG_CONV_OCC=1|2|3 to cap the
blocks per CU: how much of the distance to the MFMA peak is tile-count quantisation (partial last round)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histogan_amd import conv as C
dev = torch.device('cuda:0')
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print('HG_CONV_OCC', os.environ.get('HG_CONV_OCC', '-'))
for K, N, S in [(256, 128, 64), (128, 128, 64), (512, 256, 32), (1024, 512, 16), (128, 64, 128), (64, 32, 256)]:
    row = []
    for B in (8, 16, 24, 32, 48, 64):
        x = torch.randn(B, K, S, S, device=dev); w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
        wf = C.pack_weights(w, C.PACK_FWD)
        t = timeit(lambda: C.conv_fwd_packed(x, wf, N, 3))
        row.append(f'B={B}: {2.0 * B * S * S * K * N * 9 / t / 1e9:6.1f}')
    print(f'{K:5d}->{N:4d} @{S:3d} | ' + '  '.join(row), flush=True)
