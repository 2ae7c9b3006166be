#!/usr/bin/env python3
"""The first discriminator blocks' convolutions (16 / 32 channels on 256^2 / 128^2 maps, batch 64 = [fake; real]): times
and the bandwidth / matrix rates they correspond to."""
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
tot = 0.0
for (B, K, N, S) in [(64, 16, 16, 256), (32, 16, 16, 256), (64, 3, 16, 256), (64, 16, 32, 128), (64, 32, 32, 128), (32, 32, 3, 256)]:
    x = torch.randn(B, K, S, S, device=dev); w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
    go = torch.randn(B, N, S, S, device=dev)
    wf, wd = C.pack_weights(w, C.PACK_FWD), C.pack_weights(w, C.PACK_DGRAD)
    tf = timeit(lambda: C.conv_fwd_packed(x, wf, N, 3)); td = timeit(lambda: C.conv_dgrad_packed(go, wd, K, S, S, 3))
    fl = 2.0 * B * S * S * K * N * 9; by = 4.0 * B * S * S * (K + N)
    tot += tf + td
    print(f'{B:3d} {K:3d}->{N:3d} @{S}: fwd {tf*1e3:6.1f} us {fl/tf/1e9:5.1f} TF {by/tf/1e6:6.0f} GB/s | dgrad {td*1e3:6.1f} us {fl/td/1e9:5.1f} TF {by/td/1e6:6.0f} GB/s', flush=True)
print(f'sum {tot:.3f} ms')
