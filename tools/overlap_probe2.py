#!/usr/bin/env python3
"""Would the G phase's generator forward overlap with the D phase on a second stream?  Times the generator forward
(B = 32, no_grad) and a discriminator forward + backward (2B) alone, back to back, and on two streams."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histoGAN import Trainer  # noqa: E402

dev = torch.device('cuda:0')
tmp = tempfile.mkdtemp()
tr = Trainer('p', tmp + '/r', tmp + '/m', 256, 16, batch_size=32, hist_insz=150)
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src()
tr.train()
GAN = tr.GAN
B = 32
w = torch.randn(B, 5, 512, device=dev)
hw = torch.randn(B, 2, 512, device=dev)
noise = torch.rand(B, 256, 256, 1, device=dev)
imgs = torch.rand(2 * B, 3, 256, 256, device=dev)
side = torch.cuda.Stream(device=dev)


def gfwd():
    with torch.no_grad():
        return GAN.G(w, hw, noise)


def dstep():
    GAN.D_opt.zero_grad()
    out, _ = GAN.D(imgs)
    out.mean().backward()
    GAN._flat_d.gather()


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def serial():
    gfwd(); dstep()


def overlapped():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gfwd()
    dstep()
    torch.cuda.current_stream().wait_stream(side)


print(f'G forward alone     {timed(gfwd):.2f} ms')
print(f'D fwd+bwd (2B) alone {timed(dstep):.2f} ms')
print(f'serial              {timed(serial):.2f} ms')
print(f'two streams         {timed(overlapped):.2f} ms')
