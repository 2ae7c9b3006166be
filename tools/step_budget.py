#!/usr/bin/env python3
"""Convolution time budget of ONE HistoGAN train step (C3: 256^2, capacity 16, batch 32): every (layer, pass) with
its launches per step, HIP-event time, TFLOP/s and the time above a 130 TFLOP/s target; sorted by time.

    python tools/step_budget.py [--batch 32] [--target 130]
Passes per step (histogan_amd/trainer.py): generator forward x2 (D phase under no_grad, G phase) + data/weight
gradients x1; discriminator forward on [fake; real] (2B) + data + weight gradients, forward on B fakes + data
gradient (G phase); the gradient penalty every 4th step adds ~2 more D passes on B (counted at 1/4).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from histogan_amd import conv as C

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--target', type=float, default=130.0)
ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
dev = torch.device('cuda:0')
B = a.batch


def timeit(fn, iters=a.iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


rows = []


def layer(tag, b, K, N, S, k, stride, passes):
    """passes: dict(fwd=count, dgrad=count, wgrad=count) per step."""
    x = torch.randn(b, K, S, S, device=dev)
    w = torch.randn(N, K, k, k, device=dev) / (K * k * k) ** 0.5
    So = (S + stride - 1) // stride
    go = torch.randn(b, N, So, So, device=dev)
    wf, wd = C.pack_weights(w, C.PACK_FWD), C.pack_weights(w, C.PACK_DGRAD)
    flops = 2.0 * b * So * So * K * N * k * k
    fns = dict(fwd=lambda: C.conv_fwd_packed(x, wf, N, k, stride), dgrad=lambda: C.conv_dgrad_packed(go, wd, K, S, S, k, stride),
               wgrad=lambda: C.conv_wgrad(x, go, k, stride))
    for p, cnt in passes.items():
        if cnt:
            t = timeit(fns[p])
            rows.append(dict(tag=tag, b=b, K=K, N=N, S=S, k=k, s=stride, p=p, cnt=cnt, t=t, flops=flops))


# generator: blocks 64->2048@4 ... 64->32@256; conv1, conv2, to_rgb 1x1
filters = [64, 2048, 1024, 512, 256, 128, 64, 32]
for i in range(7):
    ci, co, S = filters[i], filters[i + 1], 4 * 2 ** i
    gp = dict(fwd=2, dgrad=1 if i else 0, wgrad=1)
    layer(f'G{i}.conv1', B, ci, co, S, 3, 1, gp)
    layer(f'G{i}.conv2', B, co, co, S, 3, 1, dict(fwd=2, dgrad=1, wgrad=1))
    layer(f'G{i}.rgb', B, co, 3, S, 1, 1, dict(fwd=2, dgrad=1, wgrad=1))
# discriminator: filters [3,16,...,2048], maps 256 ... 2
df = [3] + [16 * 2 ** i for i in range(8)]
for i in range(8):
    ci, co, S = df[i], df[i + 1], 256 // 2 ** i
    # D phase on 2B: fwd + dgrad + wgrad;  G phase on B: fwd + dgrad;  GP (1/4 of steps) on B: ~ fwd-like + dgrad-like extra
    for b, ps in ((2 * B, dict(fwd=1, dgrad=1 if i else 0, wgrad=1)), (B, dict(fwd=1 + 0.25, dgrad=1 + 0.5, wgrad=0.25))):
        layer(f'D{i}.res', b, ci, co, S, 1, 1, ps)
        layer(f'D{i}.c1', b, ci, co, S, 3, 1, ps)
        layer(f'D{i}.c2', b, co, co, S, 3, 1, dict(ps, dgrad=max(ps['dgrad'], 1) if b == 2 * B else ps['dgrad']))
        if i < 7:
            layer(f'D{i}.down', b, co, co, S, 3, 2, dict(ps, dgrad=max(ps['dgrad'], 1) if b == 2 * B else ps['dgrad']))

tot = sum(r['t'] * r['cnt'] for r in rows)
totf = sum(r['flops'] * r['cnt'] for r in rows)
print(f'conv total per step: {tot*1e3:.2f} ms, {totf/1e12:.2f} TFLOP, {totf/tot/1e12:.1f} TFLOP/s aggregate; '
      f'at {a.target:.0f} TFLOP/s: {totf/a.target/1e12*1e3:.2f} ms')
for grp in ('G', 'D'):
    for p in ('fwd', 'dgrad', 'wgrad'):
        sel = [r for r in rows if r['tag'][0] == grp and r['p'] == p]
        t = sum(r['t'] * r['cnt'] for r in sel); f = sum(r['flops'] * r['cnt'] for r in sel)
        print(f'  {grp} {p:5s}: {t*1e3:7.2f} ms  {f/t/1e12:6.1f} TFLOP/s')
print(f'{"layer":10s} {"B":>3} {"K":>5} {"N":>5} {"S":>4} k s pass  | x/step {"ms each":>8} {"TF":>6} {"ms/step":>8} {"excess":>7}')
rows.sort(key=lambda r: -(r['t'] * r['cnt'] - r['flops'] * r['cnt'] / a.target / 1e12))
for r in rows[:int(__import__('os').environ.get('HG_BUDGET_ROWS', '45'))]:
    ms = r['t'] * r['cnt'] * 1e3
    ex = ms - r['flops'] * r['cnt'] / a.target / 1e12 * 1e3
    print(f'{r["tag"]:10s} {r["b"]:3d} {r["K"]:5d} {r["N"]:5d} {r["S"]:4d} {r["k"]} {r["s"]} {r["p"]:5s} | {r["cnt"]:5.2f} '
          f'{r["t"]*1e3:8.3f} {r["flops"]/r["t"]/1e12:6.1f} {ms:8.3f} {ex:7.3f}')

blk = {}
for r in rows:
    k = r['tag'].split('.')[0]
    b = blk.setdefault(k, [0.0, 0.0])
    b[0] += r['t'] * r['cnt']; b[1] += r['flops'] * r['cnt']
print('per block: ' + '  '.join(f'{k} {v[0]*1e3:.2f} ms {v[1]/v[0]/1e12:.0f} TF' for k, v in sorted(blk.items())))
