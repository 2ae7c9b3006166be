#!/usr/bin/env python3
"""HBM-bound generator prologue / epilogue kernels at the C3 shapes: time and effective bandwidth per call.
    HG_NETS_BLOCKS=4096 python tools/nets_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histogan_amd import ops

dev = torch.device('cuda:0')
B = 32


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


tot = {}
for C_, H in ((32, 256), (64, 128), (128, 64), (256, 32), (512, 16), (1024, 8), (2048, 4)):
    conv = torch.randn(B, C_, H, H, device=dev, requires_grad=True)
    d = torch.rand(B, C_, device=dev, requires_grad=True)
    nzt = torch.rand(B, 256, 256, device=dev)
    wn = torch.randn(C_, 1, device=dev, requires_grad=True); bn = torch.randn(C_, device=dev, requires_grad=True)
    out = ops.demod_noise_lrelu(conv, d, nzt, wn, bn)
    g = torch.randn_like(out)
    t_b = timeit(lambda: torch.autograd.grad(out, (conv, d, wn, bn), g, retain_graph=True))
    nbytes = conv.numel() * 4
    x = torch.randn(B, C_, H, H, device=dev, requires_grad=True)
    s = torch.rand(B, C_, device=dev, requires_grad=True)
    y = ops.modulate(x, s, upsample=False)
    gy = torch.randn_like(y)
    t_m = timeit(lambda: torch.autograd.grad(y, (x, s), gy, retain_graph=True))
    print(f'{C_:5d} ch {H:4d}^2: dnl_bwd {t_b*1e6:8.1f} us ({4*nbytes/t_b/1e9:7.0f} GB/s of 4 tensors)   modulate_bwd {t_m*1e6:8.1f} us ({3*nbytes/t_m/1e9:7.0f} GB/s of 3 tensors)')
    tot['dnl'] = tot.get('dnl', 0) + t_b; tot['mod'] = tot.get('mod', 0) + t_m
print('sum over levels: dnl_bwd %.3f ms, modulate_bwd %.3f ms (x2 stages per level in a step)' % (tot['dnl'] * 1e3, tot['mod'] * 1e3))
