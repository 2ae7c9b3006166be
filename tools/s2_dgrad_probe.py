#!/usr/bin/env python3
"""Stride-2 data gradients of the discriminator's down-sampling convolutions at C3 (B = 64 and 32): HIP-event time,
TFLOP/s and effective GB/s (gout read + gin written), plus the max error against fp64 conv_transpose2d.

    HG_DGRAD_S2_MERGE=1 python tools/s2_dgrad_probe.py     # 1: merged launch only for small maps (round 2), 2: all maps
    HG_PARITY_XCD=0 python tools/s2_dgrad_probe.py          # class-interleaved block order instead of the XCD-paired one"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from histogan_amd import conv as C

dev = torch.device('cuda:0')
rows = []
for B in (64, 32):
    for i in range(7):
        ch, S = 16 * 2 ** i, 256 // 2 ** i
        torch.manual_seed(i)
        w = torch.randn(ch, ch, 3, 3, device=dev) / (ch * 9) ** 0.5
        go = torch.randn(B, ch, S // 2, S // 2, device=dev)
        wd = C.pack_weights(w, C.PACK_DGRAD)
        fn = lambda: C.conv_dgrad_packed(go, wd, ch, S, S, 3, 2)
        out = fn()
        err = None
        if B == 32 or S <= 64:
            ref = F.conv_transpose2d(go[:4].double(), w.double(), stride=2, padding=1, output_padding=1)
            err = float((out[:4].double() - ref).abs().max() / ref.abs().max())
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        fl = 2.0 * B * (S // 2) ** 2 * ch * ch * 9
        by = 4.0 * B * ch * (S * S + (S // 2) ** 2)
        rows.append(dict(B=B, ch=ch, S=S, us=round(t * 1e6, 1), tflops=round(fl / t / 1e12, 1), gbs=round(by / t / 1e9), err=err))
env = {k: v for k, v in os.environ.items() if k.startswith('HG_')}
print(json.dumps(dict(env=env, total_us=round(sum(r['us'] for r in rows), 1))))
for r in rows:
    print(' ', json.dumps(r))
