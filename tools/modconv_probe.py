#!/usr/bin/env python3
"""Times the fused generator stage (ops.modconv_stage fwd / bwd) against the unfused chain
(modulate -> conv2d -> demod_noise_lrelu) at the generator's layer shapes; also conv fwd / wgrad with iscale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histogan_amd import conv as C, ops
dev = torch.device('cuda:0'); B = 32

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for K, N, S in [(2048, 2048, 4), (1024, 1024, 8), (512, 512, 16), (256, 256, 32), (128, 128, 64), (64, 64, 128), (32, 32, 256)]:
    x = torch.randn(B, K, S, S, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
    s = torch.randn(B, K, device=dev) * 0.1
    go = torch.randn(B, N, S, S, device=dev)
    nzt = torch.rand(B, 256, 256, device=dev); wn = torch.randn(N, 1, device=dev); bn = torch.randn(N, device=dev)
    wf = C.pack_weights(w, C.PACK_FWD); s1 = s + 1
    t0 = timeit(lambda: C.conv_fwd_packed(x, wf, N, 3))
    t1 = timeit(lambda: C.conv_fwd_packed(x, wf, N, 3, iscale=s1))
    t2 = timeit(lambda: C.conv_wgrad(x, go, 3))
    t3 = timeit(lambda: C.conv_wgrad(x, go, 3, iscale=s1))
    xs = [t.clone().requires_grad_(True) for t in (x, s, w)]
    def unf():
        xm = ops.modulate(xs[0], xs[1], False)
        c = C.conv2d_same(xm, xs[2])
        wsq = xs[2].pow(2).sum(dim=(2, 3)); d = torch.rsqrt(torch.mm((xs[1] + 1).pow(2), wsq.t()) + 1e-8)
        return ops.demod_noise_lrelu(c, d, nzt, wn, bn)
    def fus():
        return ops.modconv_stage(xs[0], xs[1], xs[2], nzt, wn, bn, demod=True, upsample=False, act=True)
    fu, ff = timeit(lambda: unf()), timeit(lambda: fus())
    bu = timeit(lambda: torch.autograd.grad(unf(), xs, go)) - fu
    bf = timeit(lambda: torch.autograd.grad(fus(), xs, go)) - ff
    print(f'{K:5d}->{N:5d} @{S:3d} | conv fwd {t0:.3f} +iscale {t1:.3f} | wgrad {t2:.3f} +iscale {t3:.3f} | stage fwd unfused {fu:.3f} fused {ff:.3f} | bwd unfused {bu:.3f} fused {bf:.3f}', flush=True)
