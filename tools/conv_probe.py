#!/usr/bin/env python3
"""Times the implicit-GEMM convolutions (output / data-gradient / weight-gradient) at the generator's
layer shapes of the C3 config (256^2, capacity 16, batch 32) with HIP events; prints TFLOP/s per layer and
MIOpen's F.conv2d beside it.   python tools/conv_probe.py [--miopen] [--batch 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from histogan_amd import conv as C

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--miopen', action='store_true')
ap.add_argument('--iters', type=int, default=5)
args = ap.parse_args()
dev = torch.device('cuda:0')
B = args.batch
# (K, N, S, ksize) : generator convs at 256^2 / cap 16 + to-RGB 1x1 + a few D shapes
LAYERS = [(64, 2048, 4, 3), (2048, 2048, 4, 3), (2048, 1024, 8, 3), (1024, 1024, 8, 3), (1024, 512, 16, 3),
          (512, 512, 16, 3), (512, 256, 32, 3), (256, 256, 32, 3), (256, 128, 64, 3), (128, 128, 64, 3),
          (128, 64, 128, 3), (64, 64, 128, 3), (64, 32, 256, 3), (32, 32, 256, 3), (32, 3, 256, 1), (64, 3, 128, 1),
          (3, 16, 256, 3), (16, 16, 256, 3), (16, 32, 128, 3), (3, 16, 256, 1)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0, flops=0.0, mi=0.0)
print(f'{"K":>5} {"N":>5} {"S":>4} k | {"fwd ms":>8} {"TF":>6} | {"dgrad ms":>8} {"TF":>6} | {"wgrad ms":>8} {"TF":>6} | miopen fwd ms TF')
for K, N, S, k in LAYERS:
    x = torch.randn(B, K, S, S, device=dev)
    w = torch.randn(N, K, k, k, device=dev) / (K * k * k) ** 0.5
    go = torch.randn(B, N, S, S, device=dev)
    wf, wd = C.pack_weights(w, C.PACK_FWD), C.pack_weights(w, C.PACK_DGRAD)
    flops = 2.0 * B * S * S * K * N * k * k
    tf = timeit(lambda: C.conv_fwd_packed(x, wf, N, k), args.iters)
    td = timeit(lambda: C.conv_dgrad_packed(go, wd, K, S, S, k), args.iters)
    tw = timeit(lambda: C.conv_wgrad(x, go, k), args.iters)
    tp = timeit(lambda: C.pack_weights(w, C.PACK_FWD), args.iters)
    line = (f'{K:5d} {N:5d} {S:4d} {k} | {tf*1e3:8.3f} {flops/tf/1e12:6.1f} | {td*1e3:8.3f} {flops/td/1e12:6.1f} | '
            f'{tw*1e3:8.3f} {flops/tw/1e12:6.1f} | pack {tp*1e3:.3f}')
    if args.miopen:
        tm = timeit(lambda: F.conv2d(x, w, padding=k // 2), args.iters)
        line += f' | {tm*1e3:8.3f} {flops/tm/1e12:6.1f}'
        tot['mi'] += tm
    print(line, flush=True)
    if k == 3 and K >= 32:
        tot['fwd'] += tf; tot['dgrad'] += td; tot['wgrad'] += tw; tot['flops'] += flops
print('generator 3x3 layers: fwd %.2f ms (%.1f TF)  dgrad %.2f ms (%.1f TF)  wgrad %.2f ms (%.1f TF)' % (
    tot['fwd'] * 1e3, tot['flops'] / tot['fwd'] / 1e12, tot['dgrad'] * 1e3, tot['flops'] / tot['dgrad'] / 1e12,
    tot['wgrad'] * 1e3, tot['flops'] / tot['wgrad'] / 1e12))

# ---- discriminator shapes (256^2, cap 16): (K, N, S_in, ksize, stride)
print('\ndiscriminator layers')
DL = []
f = [3, 16, 32, 64, 128, 256, 512, 1024, 2048]
for i in range(8):
    S = 256 >> i
    DL += [(f[i], f[i + 1], S, 1, 1), (f[i], f[i + 1], S, 3, 1), (f[i + 1], f[i + 1], S, 3, 1)]
    if i < 7:
        DL.append((f[i + 1], f[i + 1], S, 3, 2))
tt = dict(f=0.0, d=0.0, w=0.0, fl=0.0)
for K, N, S, k, st in DL:
    x = torch.randn(B, K, S, S, device=dev)
    w = torch.randn(N, K, k, k, device=dev) / (K * k * k) ** 0.5
    So = (S - 1) // st + 1
    go = torch.randn(B, N, So, So, device=dev)
    wf, wd = C.pack_weights(w, C.PACK_FWD), C.pack_weights(w, C.PACK_DGRAD)
    flops = 2.0 * B * So * So * K * N * k * k
    tf = timeit(lambda: C.conv_fwd_packed(x, wf, N, k, st), args.iters)
    td = timeit(lambda: C.conv_dgrad_packed(go, wd, K, S, S, k, st), args.iters)
    tw = timeit(lambda: C.conv_wgrad(x, go, k, st), args.iters)
    print(f'{K:5d} {N:5d} {S:4d} k{k} s{st} | fwd {tf*1e3:7.3f} ms {flops/tf/1e12:6.1f} TF | dgrad {td*1e3:7.3f} {flops/td/1e12:6.1f} | '
          f'wgrad {tw*1e3:7.3f} {flops/tw/1e12:6.1f}', flush=True)
    tt['f'] += tf; tt['d'] += td; tt['w'] += tw; tt['fl'] += flops
print('discriminator: fwd %.2f ms (%.1f TF)  dgrad %.2f ms (%.1f TF)  wgrad %.2f ms (%.1f TF)' % (
    tt['f'] * 1e3, tt['fl'] / tt['f'] / 1e12, tt['d'] * 1e3, tt['fl'] / tt['d'] / 1e12, tt['w'] * 1e3, tt['fl'] / tt['w'] / 1e12))
