#!/usr/bin/env python3
"""configs[1] histogram launches for rocprofv3 (kernel trace / PMC passes): N x (forward + backward) at 32x3x256^2, h = 64.
    HG_HIST_ITERS=40 HG_HIST_METHOD=inverse-quadratic HG_HIST_H=64 HG_HIST_INSZ=256 python tools/hist_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histogan_amd.hist import HistConfig, rgbuv_hist  # noqa: E402

dev = torch.device('cuda:0')
n = int(os.environ.get('HG_HIST_ITERS', '40'))
B = int(os.environ.get('HG_HIST_B', '32'))
S = int(os.environ.get('HG_HIST_S', '256'))
h = int(os.environ.get('HG_HIST_H', '64'))
cfg = HistConfig(h=h, insz=int(os.environ.get('HG_HIST_INSZ', str(S))), method=os.environ.get('HG_HIST_METHOD', 'inverse-quadratic'),
                 sigma=0.02, hist_boundary=[float(v) for v in os.environ['HG_HIST_BOUNDARY'].split(',')] if 'HG_HIST_BOUNDARY' in os.environ else None)
x = torch.rand(B, 3, S, S, device=dev).requires_grad_(True)
go = torch.randn(B, 3, h, h, device=dev)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for i in range(n + 3):
    x.grad = None
    e[0].record(); o = rgbuv_hist(x, cfg); e[1].record(); o.backward(go); e[2].record()
    torch.cuda.synchronize()
    if i >= 3:
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
print(f'{cfg.method} h={h} B={B} {S}^2->insz {cfg.insz}: fwd {tf/n*1e3:.1f} us  bwd {tb/n*1e3:.1f} us (event-timed, incl. helpers)')
