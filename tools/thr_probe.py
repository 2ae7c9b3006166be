#!/usr/bin/env python3
"""Thresholding histogram (scatter-add kernels) at configs[1] (32x3x256x256, h=64): kernel times (HIP events) and the
algorithmic HBM rate.   python tools/thr_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histogan_amd.hist import HistConfig, rgbuv_hist  # noqa: E402

dev = torch.device('cuda:0')
B, S, h = 32, 256, 64
for name, x in (('uniform', torch.rand(B, 3, S, S, device=dev)), ('constant colour', torch.full((B, 3, S, S), 0.4, device=dev))):
    for method in ('thresholding', 'RBF', 'inverse-quadratic'):
        cfg = HistConfig(h=h, insz=S, method=method, sigma=0.02)
        xg = x.clone().requires_grad_(True)
        out = rgbuv_hist(xg, cfg)
        go = torch.randn_like(out)
        for _ in range(3):
            xg.grad = None
            rgbuv_hist(xg, cfg).backward(go)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 20
        tf = tb = 0.0
        for _ in range(n):
            xg.grad = None
            e[0].record(); o = rgbuv_hist(xg, cfg); e[1].record(); o.backward(go); e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        tf, tb = tf / n * 1e-3, tb / n * 1e-3
        bf, bb = B * (3 * S * S + 3 * h * h) * 4, B * (6 * S * S + 3 * h * h) * 4
        print(f'{name:16s} {method:18s} fwd {tf*1e6:8.1f} us ({bf/tf/1e9:7.1f} GB/s)  bwd {tb*1e6:8.1f} us ({bb/tb/1e9:7.1f} GB/s)  '
              f'fwd+bwd {(bf+bb)/(tf+tb)/1e9:7.1f} GB/s = {(bf+bb)/(tf+tb)/8e12*100:5.1f} % of 8 TB/s')
