#!/usr/bin/env python3
"""Times hg_wino_conv2d on four layers for the current library (HG_LIB_TAG): used with the HG_WINO_DBG ablation builds
(results of those builds are garbage by construction; only the times mean anything).  The -DHG_WINO_DBG=... switches were
removed from hg_wino.hip after the measurement (profiles/r05_wino_ablation.txt); they are in the history at commit f419444
(`git show f419444:histogan_amd/csrc/hg_wino.hip`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from histogan_amd import conv as C
dev = torch.device('cuda:0')
B = 32
out = []
for K, N, S in ((256, 128, 64), (512, 512, 16), (1024, 1024, 8), (64, 64, 128)):
    x = torch.randn(B, K, S, S, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
    u = C._wino_pack(w, C.PACK_FWD)
    for _ in range(3):
        C.wino_conv(x, u, N)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            C.wino_conv(x, u, N)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    out.append('%d->%d@%d %.3f ms' % (K, N, S, min(ts)))
print(os.environ.get('HG_LIB_TAG', 'default'), ' | '.join(out), flush=True)
