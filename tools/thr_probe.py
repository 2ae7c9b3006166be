#!/usr/bin/env python3
"""HBM-side histogram methods (scatter-add kernels) at configs[1] (Bx3x256x256, h=64) through the C ABI with
preallocated buffers (as bench.py times them): HIP-event times of the forward / backward call and the algorithmic
HBM rate.   python tools/thr_probe.py [B ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histogan_amd import hist as HH  # noqa: E402
from histogan_amd._lib import check, lib  # noqa: E402

dev = torch.device('cuda:0')
S, h = 256, 64
for B in [int(a) for a in sys.argv[1:]] or [32, 256]:
    for name, x in (('uniform', torch.rand(B, 3, S, S, device=dev)), ('constant colour', torch.full((B, 3, S, S), 0.4, device=dev))):
        for method in ('thresholding', 'RBF'):
            cfg = HH.HistConfig(h=h, insz=S, method=method, sigma=0.02)
            p, keep = HH._make_params(x, cfg)
            fb, bb = HH._ws_bytes(p)
            out = torch.empty(B, 3, h, h, device=dev); sums = torch.empty(B, device=dev); gx = torch.empty_like(x)
            gout = torch.rand(B, 3, h, h, device=dev) - 0.5
            ws = torch.empty(max(fb, bb, 4), dtype=torch.uint8, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            n = int(os.environ.get('HG_THR_ITERS', '30'))
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
            for it in range(n + 3):
                e = evs[max(it - 3, 0)]
                e[0].record()
                check(lib.hg_rgbuv_hist_fwd(ctypes.byref(p), x.data_ptr(), out.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel(), st), 'fwd')
                e[1].record()
                check(lib.hg_rgbuv_hist_bwd(ctypes.byref(p), x.data_ptr(), gout.data_ptr(), out.data_ptr(), sums.data_ptr(), gx.data_ptr(), ws.data_ptr(), ws.numel(), st), 'bwd')
                e[2].record()
            torch.cuda.synchronize()
            tf = sum(e[0].elapsed_time(e[1]) for e in evs) / n * 1e-3
            tb = sum(e[1].elapsed_time(e[2]) for e in evs) / n * 1e-3
            bf, bbw = B * (3 * S * S + 3 * h * h) * 4, B * (6 * S * S + 3 * h * h) * 4
            print(f'B={B:3d} {name:16s} {method:13s} fwd {tf*1e6:7.1f} us ({bf/tf/1e9:7.1f} GB/s)  bwd {tb*1e6:7.1f} us ({bbw/tb/1e9:7.1f} GB/s)  '
                  f'fwd+bwd {(bf+bbw)/(tf+tb)/1e9:7.1f} GB/s = {(bf+bbw)/(tf+tb)/8e12*100:5.1f} % of 8 TB/s', flush=True)
