#!/usr/bin/env python3
"""Do an MFMA-bound weight-gradient kernel and an HBM-bound elementwise kernel overlap on two HIP streams?
Times: each alone, back to back on one stream, concurrently on two streams.   python tools/overlap_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histogan_amd import conv as C  # noqa: E402

dev = torch.device('cuda:0')
B, K, N, S = 32, 256, 128, 64
x = torch.randn(B, K, S, S, device=dev)
go = torch.randn(B, N, S, S, device=dev)
big = torch.randn(32, 64, 128, 128, device=dev)          # 134 MB: the elementwise work of a backward stage
big2 = torch.randn_like(big)
side = torch.cuda.Stream(device=dev)


def wg():
    return C.conv_wgrad(x, go, 3)


def ew():
    a = torch.nn.functional.leaky_relu(big, 0.2)
    return a * big2 + big


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def both_serial():
    wg(); ew()


def both_overlap():
    ev = torch.cuda.current_stream().record_event()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        wg()
    ew()
    torch.cuda.current_stream().wait_stream(side)


print(f'wgrad alone      {timed(wg):.3f} ms')
print(f'elementwise alone {timed(ew):.3f} ms')
print(f'serial            {timed(both_serial):.3f} ms')
print(f'two streams       {timed(both_overlap):.3f} ms')
dg_w = C.pack_weights(torch.randn(N, K, 3, 3, device=dev), C.PACK_DGRAD)


def dg():
    return C.conv_dgrad_packed(go, dg_w, K, S, S, 3)


def conv_overlap():
    ev = torch.cuda.current_stream().record_event()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        wg()
    dg()
    torch.cuda.current_stream().wait_stream(side)


print(f'dgrad alone       {timed(dg):.3f} ms')
print(f'dgrad+wgrad serial {timed(lambda: (dg(), wg())):.3f} ms')
print(f'dgrad || wgrad    {timed(conv_overlap):.3f} ms')
