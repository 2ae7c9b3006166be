#!/usr/bin/env python3
"""Golden vectors for the network part of the path, from the UNMODIFIED reference classes.

    python tests/golden/make_golden_nets.py      # writes tests/golden/nets_small.npz

histoGAN/histoGAN.py hard-imports packages that are absent here (torch_optimizer, torchvision,
vector_quantize_pytorch, linear_attention_transformer) and asserts CUDA at import; they are stubbed
in sys.modules and torch.cuda.is_available is patched for the duration of the import only, so the
reference's own Conv2DMod / Generator / Discriminator / vectorizers / gradient_penalty run on CPU
(SURVEY.md section 8c).  gradient_penalty's `.cuda()` is patched to a no-op for the CPU run.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ('torch_optimizer', 'torchvision', 'torchvision.transforms', 'vector_quantize_pytorch',
                 'linear_attention_transformer', 'retry', 'retry.api'):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules['torch_optimizer'].DiffGrad = object
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['vector_quantize_pytorch'].VectorQuantize = object
    sys.modules['linear_attention_transformer'].ImageLinearAttention = object
    sys.path.insert(0, REF)
    real = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        import histoGAN.histoGAN as R
    finally:
        torch.cuda.is_available = real
        sys.path.pop(0)
    return R


def main():
    R = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self       # CPU run of gradient_penalty / helpers
    torch.manual_seed(0)
    S_, CAP, LAT, HB, B = 32, 4, 32, 16, 2
    G = R.Generator(S_, LAT, network_capacity=CAP)
    D = R.Discriminator(S_, network_capacity=CAP)
    SV = R.StyleVectorizer(LAT, 3)
    HV = R.HistVectorizer(HB, LAT, 3)
    # non-zero noise weights so the noise path (H<->W swapped permute) is exercised
    for blk in G.blocks:
        for lin in (blk.to_noise1, blk.to_noise2):
            torch.nn.init.normal_(lin.weight, std=0.5)
            torch.nn.init.normal_(lin.bias, std=0.1)
    L = G.num_layers
    out = {}

    def put(prefix, sd):
        for k, v in sd.items():
            out[f'{prefix}/{k}'] = v.detach().numpy()

    put('G', G.state_dict()); put('D', D.state_dict()); put('S', SV.state_dict()); put('H', HV.state_dict())

    # Conv2DMod alone (demod on and off, 3x3 and 1x1)
    for tag, (ci, co, k, demod) in dict(c3=(8, 12, 3, True), c1=(8, 3, 1, False)).items():
        conv = R.Conv2DMod(ci, co, k, demod=demod)
        x = torch.randn(B, ci, 8, 8, requires_grad=True)
        y = torch.randn(B, ci, requires_grad=True)
        o = conv(x, y)
        go = torch.randn_like(o)
        gx, gy, gw = torch.autograd.grad(o, (x, y, conv.weight), go)
        out.update({f'{tag}/weight': conv.weight.detach().numpy(), f'{tag}/x': x.detach().numpy(),
                    f'{tag}/y': y.detach().numpy(), f'{tag}/out': o.detach().numpy(), f'{tag}/go': go.numpy(),
                    f'{tag}/gx': gx.numpy(), f'{tag}/gy': gy.numpy(), f'{tag}/gw': gw.numpy()})

    # vectorizers
    z = torch.randn(B, LAT)
    hist = torch.rand(B, 3, HB, HB); hist = hist / hist.sum(dim=(1, 2, 3), keepdim=True)
    w = SV(z); hw = HV(hist)
    out.update(z=z.numpy(), hist=hist.numpy(), w=w.detach().numpy(), hw=hw.detach().numpy())

    # generator forward / backward
    styles = torch.randn(B, L - 2, LAT, requires_grad=True)
    hists = torch.randn(B, 2, LAT, requires_grad=True)
    noise = torch.rand(B, S_, S_, 1)
    rgb = G(styles, hists, noise)
    go = torch.randn_like(rgb)
    params = dict(G.named_parameters())
    names = ['initial_block', 'blocks.0.conv1.weight', 'blocks.1.to_noise1.weight', 'blocks.1.to_noise2.bias',
             'blocks.2.to_style1.weight', 'blocks.3.to_rgb.conv.weight', 'blocks.3.conv2.weight']
    grads = torch.autograd.grad(rgb, [styles, hists] + [params[n] for n in names], go)
    out.update(g_styles=styles.detach().numpy(), g_hists=hists.detach().numpy(), g_noise=noise.numpy(),
               g_rgb=rgb.detach().numpy(), g_go=go.numpy(), g_grad_styles=grads[0].numpy(),
               g_grad_hists=grads[1].numpy())
    for n, g in zip(names, grads[2:]):
        out[f'g_grad/{n}'] = g.numpy()

    # discriminator forward, gradient penalty (double backward) and its gradient w.r.t. a few params
    img = torch.rand(B, 3, S_, S_, requires_grad=True)
    logits, q = D(img)
    gp = R.gradient_penalty(img, logits)
    dloss = (torch.nn.functional.relu(1 + logits)).mean() + gp
    dparams = dict(D.named_parameters())
    dnames = ['blocks.0.conv_res.weight', 'blocks.0.net.0.bias', 'blocks.2.downsample.weight', 'to_logit.weight']
    dgr = torch.autograd.grad(dloss, [dparams[n] for n in dnames])
    out.update(d_img=img.detach().numpy(), d_logits=logits.detach().numpy(), d_gp=np.float64(gp.item()),
               d_loss=np.float64(dloss.item()))
    for n, g in zip(dnames, dgr):
        out[f'd_grad/{n}'] = g.numpy()

    # helpers with fixed inputs
    a = torch.randn(B, LAT); b2 = torch.randn(B, LAT)
    out['styles_def'] = R.styles_def_to_tensor([(a, 2), (b2, L - 4)]).numpy()
    out['styles_def_a'] = a.numpy(); out['styles_def_b'] = b2.numpy()
    out['meta'] = np.array([S_, CAP, LAT, HB, B, L])
    np.savez_compressed(os.path.join(HERE, 'nets_small.npz'), **out)
    print('wrote nets_small.npz with', len(out), 'arrays; G params',
          sum(p.numel() for p in G.parameters()), 'D params', sum(p.numel() for p in D.parameters()))


if __name__ == '__main__':
    main()
