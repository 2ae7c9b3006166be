#!/usr/bin/env python3
"""Debug: whole-discriminator gradient penalty with the trainer's initialisation (kaiming_normal) vs the fp64 oracle,
outside the Trainer; then with the first-order gradient compared per block."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histoGAN import Discriminator  # noqa: E402
from histoGAN.histoGAN import gradient_penalty  # noqa: E402
from oracle import histogan_nets as N  # noqa: E402

dev = torch.device('cuda:0')
rel = lambda a, t: float((a.double() - t.double()).abs().max() / t.double().abs().max().clamp_min(1e-300))
for init in ('default', 'kaiming_normal'):
    torch.manual_seed(22)
    D = Discriminator(256, network_capacity=16).to(dev)
    if init == 'kaiming_normal':
        for m in D.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
                torch.nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')
    img = torch.rand(2, 3, 256, 256, device=dev)
    x = img.clone().requires_grad_(True)
    logits, _ = D(x)
    gp = gradient_penalty(x, logits)
    names = [n for n, p in D.named_parameters() if p.dim() == 4]
    params = dict(D.named_parameters())
    grads = torch.autograd.grad(gp, [params[n] for n in names], retain_graph=True)
    gx1, = torch.autograd.grad(logits.sum(), x, retain_graph=True)
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    xc = img.double().clone().requires_grad_(True)
    lo = N.discriminator(sd, xc, len(D.blocks))
    gpo = N.gradient_penalty(xc, lo)
    gr = torch.autograd.grad(gpo, [sd[n] for n in names], retain_graph=True)
    gx1r, = torch.autograd.grad(lo.sum(), xc, retain_graph=True)
    rows = sorted(((rel(a, b), n) for n, a, b in zip(names, grads, gr)), reverse=True)
    print(init, 'logits', rel(logits.detach(), lo.detach()), 'gp', float(gp), float(gpo), 'first-order gx', rel(gx1, gx1r))
    print('   worst:', [(f'{e:.1e}', n) for e, n in rows[:6]])
    # first-order gradient per channel / pixel parity of the input image gradient
    e = (gx1.double() - gx1r).abs()
    print('   gx err by parity (y%2, x%2):', [[float(e[:, :, py::2, px::2].max()) for px in (0, 1)] for py in (0, 1)], 'scale', float(gx1r.abs().max()))
