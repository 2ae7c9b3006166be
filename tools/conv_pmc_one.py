#!/usr/bin/env python3
"""bench.py's roofline launch in isolation (hg_conv2d_fwd, 256->128 ch, 64x64, batch 32; + wgrad, + the hist kernels
at configs[1]) for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histogan_amd import conv as C
from histogan_amd.hist import HistConfig, hellinger_loss, rgbuv_hist
dev = torch.device('cuda:0')
B, K, N, S = 32, 256, 128, 64
x = torch.randn(B, K, S, S, device=dev)
w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
go = torch.randn(B, N, S, S, device=dev)
wf = C.pack_weights(w, C.PACK_FWD)
for _ in range(int(os.environ.get('HG_ONE_ITERS', 4))):     # 4 for PMC passes; 60 for a steady-state kernel trace
    C.conv_fwd_packed(x, wf, N, 3)
    C.conv_wgrad(x, go, 3)
cfg = HistConfig(h=64, insz=256, method='inverse-quadratic', sigma=0.02)
xi = torch.rand(32, 3, 256, 256, device=dev, requires_grad=True)
tg = rgbuv_hist(torch.rand(32, 3, 256, 256, device=dev), cfg).detach()
for _ in range(3):
    xi.grad = None
    hellinger_loss(tg, rgbuv_hist(xi, cfg), alpha=2.0).backward()
torch.cuda.synchronize()
