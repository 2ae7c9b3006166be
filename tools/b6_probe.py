#!/usr/bin/env python3
"""k_conv_b6 (bf16x6) vs k_conv (fp32 MFMA): time and error against fp64 at the generator's layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from histogan_amd import conv as C
dev = torch.device('cuda:0'); B = 32
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
for K, N, S in [(2048, 1024, 8), (1024, 512, 16), (512, 512, 16), (512, 256, 32), (256, 256, 32), (256, 128, 64), (128, 128, 64), (128, 64, 128), (64, 64, 128), (64, 32, 256), (32, 32, 256)]:
    x = torch.randn(B, K, S, S, device=dev); w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
    wf, w6 = C.pack_weights(w, C.PACK_FWD), C.pack_b6(w, C.PACK_FWD)
    t32 = timeit(lambda: C.conv_fwd_packed(x, wf, N, 3)); t6 = timeit(lambda: C.conv_b6(x, w6, N, nprod=6)); t9 = timeit(lambda: C.conv_b6(x, w6, N, nprod=9))
    tp = timeit(lambda: C.pack_b6(w, C.PACK_FWD))
    fl = 2.0 * B * S * S * K * N * 9
    err = ''
    if S <= 32:
        ref = F.conv2d(x[:4].double(), w.double(), padding=1)
        err = f' err vs fp64: f32 {rel(C.conv_fwd_packed(x[:4].contiguous(), wf, N, 3), ref):.1e}  b6 {rel(C.conv_b6(x[:4].contiguous(), w6, N, nprod=6), ref):.1e}  b9 {rel(C.conv_b6(x[:4].contiguous(), w6, N, nprod=9), ref):.1e}'
    print(f'{K:5d}->{N:5d} @{S:3d} | f32 {t32:.3f} ms {fl/t32/1e9:6.1f} TF | b6 {t6:.3f} ms {fl/t6/1e9:6.1f} | b9 {t9:.3f} ms {fl/t9/1e9:6.1f} TF-equiv | pack {tp:.3f} ms{err}', flush=True)
