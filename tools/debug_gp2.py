#!/usr/bin/env python3
"""Debug: double backward of ONE discriminator block (trainer-style kaiming_normal init) vs fp64 torch, per block shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from histogan_amd.nets import DiscriminatorBlock  # noqa: E402
from histogan_amd.conv import input_grads_only  # noqa: E402

dev = torch.device('cuda:0')
rel = lambda a, t: float((a.double() - t.double()).abs().max() / t.double().abs().max().clamp_min(1e-300))


def ref_block(sd, x, down):
    res = F.conv2d(x, sd['conv_res.weight'], sd['conv_res.bias'])
    h = F.leaky_relu(F.conv2d(x, sd['net.0.weight'], sd['net.0.bias'], padding=1), 0.2)
    h = F.leaky_relu(F.conv2d(h, sd['net.2.weight'], sd['net.2.bias'], padding=1), 0.2)
    y = h + res
    if down:
        y = F.conv2d(y, sd['downsample.weight'], sd['downsample.bias'], padding=1, stride=2)
    return y


for (ci, co, S, B, scale) in [(64, 128, 32, 2, 1.0), (64, 128, 32, 2, 30.0), (128, 256, 16, 2, 30.0), (32, 64, 64, 2, 30.0), (64, 128, 32, 8, 30.0)]:
    torch.manual_seed(5)
    blk = DiscriminatorBlock(ci, co, downsample=True).to(dev)
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')
    x = (torch.randn(B, ci, S, S, device=dev) * scale).requires_grad_(True)
    go = torch.randn(B, co, S // 2, S // 2, device=dev)
    names = [n for n, _ in blk.named_parameters()]
    params = dict(blk.named_parameters())

    def second(fn, x, go, plist, ctx):
        y = fn(x)
        with ctx():
            gx, = torch.autograd.grad(y, x, go, create_graph=True)
        pen = ((gx.reshape(gx.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean() * 10
        return y.detach(), gx.detach(), torch.autograd.grad(pen, plist, allow_unused=True)

    import contextlib
    y, gx, gr = second(blk, x, go, [params[n] for n in names], input_grads_only)
    sd = {k: v.detach().double().requires_grad_(True) for k, v in blk.state_dict().items()}
    xd = x.detach().double().requires_grad_(True)
    yr, gxr, grr = second(lambda t: ref_block(sd, t, True), xd, go.double(), [sd[n] for n in names], contextlib.nullcontext)
    print(f'block {ci}->{co} @{S} B={B} scale {scale}: y {rel(y, yr):.2e}  gx {rel(gx, gxr):.2e}   ' +
          '  '.join(f'{n.replace(".weight", ".w").replace(".bias", ".b")} {rel(a, b):.1e}' for n, a, b in zip(names, gr, grr) if a is not None and b is not None))
    w = max([t for t in zip(names, gr, grr) if t[1] is not None and t[2] is not None], key=lambda t: rel(t[1], t[2]))
    e = (w[1].double() - w[2]).abs()
    idx = (e > 0.3 * e.max()).nonzero()
    print('   worst', w[0], 'n elements > 30% max err:', idx.shape[0], idx[:8].tolist())
