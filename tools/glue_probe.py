#!/usr/bin/env python3
"""Host-side cost of one autograd-wrapped kernel call (tiny tensors: the GPU work is negligible)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histogan_amd import ops  # noqa: E402
from histogan_amd.conv import conv2d, conv2d_lrelu, enable_pack_cache  # noqa: E402

dev = torch.device('cuda:0')
x = torch.randn(2, 16, 8, 8, device=dev, requires_grad=True)
w = torch.nn.Parameter(torch.randn(16, 16, 3, 3, device=dev))
b = torch.nn.Parameter(torch.randn(16, device=dev))
enable_pack_cache([w])
s = torch.randn(2, 16, device=dev, requires_grad=True)


def bench(name, fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'{name:34s} {1e6*(t1-t0)/n:7.1f} us/call (host)')


bench('conv2d fwd (no grad)', lambda: conv2d(x.detach(), w.detach(), b.detach()))
bench('conv2d fwd (grad graph)', lambda: conv2d(x, w, b))
bench('conv2d_lrelu fwd (grad graph)', lambda: conv2d_lrelu(x, w, b))
bench('conv2d fwd+bwd', lambda: conv2d(x, w, b).sum().backward())
bench('modulate fwd (grad graph)', lambda: ops.modulate(x, s))
bench('torch add', lambda: x + x)
bench('torch F.conv2d fwd (grad graph)', lambda: torch.nn.functional.conv2d(x, w, b, padding=1))
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    conv2d(x, w, b)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
