#!/usr/bin/env python3
"""Small-map convolutions of the discriminator's deep blocks (few pixels, many channels): forward / data gradient /
weight gradient times (HIP events, 20 launches)."""
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
print('HG_CONV_SPLIT_TARGET', os.environ.get('HG_CONV_SPLIT_TARGET'))
tot = 0.0
for (B, K, N, S, k, st) in [(64, 1024, 1024, 4, 3, 2), (32, 1024, 1024, 4, 3, 2), (64, 512, 512, 8, 3, 2), (32, 512, 512, 8, 3, 2), (64, 256, 256, 16, 3, 2),
                            (64, 2048, 2048, 2, 3, 1), (64, 1024, 2048, 2, 3, 1), (64, 1024, 1024, 4, 3, 1), (64, 512, 1024, 4, 3, 1), (64, 1024, 2048, 2, 1, 1)]:
    x = torch.randn(B, K, S, S, device=dev); So = (S + st - 1) // st
    w = torch.randn(N, K, k, k, device=dev) / (K * k * k) ** 0.5
    go = torch.randn(B, N, So, So, device=dev)
    wf, wd = C.pack_weights(w, C.PACK_FWD), C.pack_weights(w, C.PACK_DGRAD)
    tf = timeit(lambda: C.conv_fwd_packed(x, wf, N, k, st)); td = timeit(lambda: C.conv_dgrad_packed(go, wd, K, S, S, k, st)); tw = timeit(lambda: C.conv_wgrad(x, go, k, st))
    fl = 2.0 * B * So * So * K * N * k * k
    tot += tf + td + tw
    print(f'{B:3d} {K:5d}->{N:5d} @{S} k{k} s{st}: fwd {tf*1e3:6.1f} us {fl/tf/1e9:6.1f} TF | dgrad {td*1e3:6.1f} us {fl/td/1e9:6.1f} | wgrad {tw*1e3:6.1f} us {fl/tw/1e9:6.1f}', flush=True)
print(f'sum {tot:.3f} ms')
