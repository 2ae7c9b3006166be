#!/usr/bin/env python3
"""Golden vectors for the networks AT THE BENCH WIDTH (BASELINE configs[2]: 256^2, network_capacity 16, latent 512), from
the UNMODIFIED reference classes -- pins oracle/histogan_nets.py where the GPU parity tests of tests/test_c3_parity_gpu.py
use it (nets_small.npz pins it at 32^2 / capacity 4 only).

    python tests/golden/make_golden_nets_c3.py      # ~2 min on 8 CPU threads; writes tests/golden/nets_c3.npz (~1 MB)

The reference's Generator(256, 512, 16) / Discriminator(256, 16) (histoGAN/histoGAN.py:529-631) hold 83 M + 91 M
parameters -- too many to store -- so the weights are NOT stored: `synth_state_dict` below fills every tensor of the
reference's state_dict from a seeded torch.Generator (normal, std = sqrt(2 / fan_in) for weights as the reference's
kaiming init :684-690, N(0, 1) initial block, non-zero noise layers and biases so that every path carries signal; the
logit layer scaled by 1e-3 so that |logit| < 1 and the hinge is active).  The file keeps the (name, shape) list, so a test
rebuilds the identical state dict from the seed alone (plus a fingerprint of a few tensors that guards the generator's
reproducibility).  Stored from the reference run at B = 1: rgb (3 x 256 x 256), logits, gradient penalty, hinge + penalty
loss, every gradient tensor with <= 16 384 elements in full, and for EVERY parameter gradient two fp64 reductions
(the dot product with a seeded +-1 vector and the L2 norm) -- a checksum that pins the large tensors too.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

SEED = 4242
FULL_MAX = 16384


def synth_state_dict(spec, seed=SEED, dtype=torch.float32):
    """spec: [(name, shape)] in state_dict order -> {name: tensor}.  Deterministic in (spec, seed): one CPU generator,
    tensors drawn in order."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    sd = {}
    for name, shape in spec:
        shape = tuple(shape)
        t = torch.randn(shape, generator=g, dtype=torch.float32)
        if name == 'initial_block':
            std = 1.0
        elif 'to_noise' in name:
            std = 0.3 if name.endswith('weight') else 0.1
        elif name.endswith('bias'):
            std = 0.1
        else:
            fan_in = int(np.prod(shape[1:]))
            std = (2.0 / fan_in) ** 0.5
            if name.startswith('to_logit'):
                std *= 1e-3
        sd[name] = (t * std).to(dtype)
    return sd


def sign_vector(n, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randint(0, 2, (n,), generator=g, dtype=torch.int8).double() * 2 - 1)


def reductions(t, seed):
    t = t.detach().double().flatten()
    return np.array([float(t @ sign_vector(t.numel(), seed)), float(t.norm())])


def fingerprint(sd):
    return np.array([float(v.double().sum()) for k, v in list(sd.items())[::7]])


def main():
    from make_golden_nets import import_reference
    R = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    S_, CAP, LAT, B = 256, 16, 512, 1
    torch.manual_seed(0)
    G = R.Generator(S_, LAT, network_capacity=CAP)
    D = R.Discriminator(S_, network_capacity=CAP)
    L = G.num_layers
    out = {}
    specs = {}
    for tag, net in (('G', G), ('D', D)):
        spec = [(k, list(v.shape)) for k, v in net.state_dict().items()]
        specs[tag] = spec
        sd = synth_state_dict(spec, SEED + (0 if tag == 'G' else 1))
        net.load_state_dict(sd, strict=True)
        out[f'{tag}_fingerprint'] = fingerprint(sd)
    out['spec'] = np.array(json.dumps(specs))

    g = torch.Generator(device='cpu').manual_seed(SEED + 2)
    styles = torch.randn(B, L - 2, LAT, generator=g).requires_grad_(True)
    hists = torch.randn(B, 2, LAT, generator=g).requires_grad_(True)
    noise = torch.rand(B, S_, S_, 1, generator=g)
    go = torch.randn(B, 3, S_, S_, generator=g)
    # The discriminator's image: the first seed whose LeakyReLU pre-activations all stay clear of zero in fp64 (relative to
    # the layer maximum).  A pre-activation within fp32 rounding of zero takes the other slope in any two fp32 evaluations
    # that round differently (this CPU run, a GPU kernel), and through the gradient penalty's second-order terms ONE such
    # pixel moves a weight gradient of its layer by 1e-3 (measured with the first image tried: blocks.1.net.2, 8.6e-4) --
    # a property of fp32 at a kink, not of either implementation (tests/oracle_step.lrelu_margin).
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle_step import lrelu_margin
    sd_d = {k: v.detach() for k, v in D.state_dict().items()}
    for img_seed in range(SEED + 3, SEED + 63):
        img = torch.rand(B, 3, S_, S_, generator=torch.Generator(device='cpu').manual_seed(img_seed))
        margin = lrelu_margin(sd_d, img, L + 1)
        print('image seed', img_seed, 'LeakyReLU margin', margin)
        if margin > 2e-7:
            break
    else:
        raise SystemExit('no image with a clear LeakyReLU margin found')
    out['img_seed'], out['img_margin'] = np.int64(img_seed), np.float64(margin)

    rgb = G(styles, hists, noise)
    names = [n for n, _ in G.named_parameters()]
    params = dict(G.named_parameters())
    grads = torch.autograd.grad(rgb, [styles, hists] + [params[n] for n in names], go)
    out.update(g_rgb=rgb.detach().numpy(), g_grad_styles=grads[0].numpy(), g_grad_hists=grads[1].numpy())
    for i, (n, gr) in enumerate(zip(names, grads[2:])):
        out[f'g_red/{n}'] = reductions(gr, SEED + 100 + i)
        if gr.numel() <= FULL_MAX:
            out[f'g_grad/{n}'] = gr.numpy()
    print('generator done: rgb max', float(rgb.abs().max()))

    x = img.clone().requires_grad_(True)
    logits, _ = D(x)
    gp = R.gradient_penalty(x, logits)
    loss = torch.nn.functional.relu(1 + logits).mean() + gp
    dnames = [n for n, _ in D.named_parameters()]
    dparams = dict(D.named_parameters())
    dgr = torch.autograd.grad(loss, [dparams[n] for n in dnames])
    out.update(d_logits=logits.detach().numpy().reshape(-1), d_gp=np.float64(gp.item()), d_loss=np.float64(loss.item()))
    for i, (n, gr) in enumerate(zip(dnames, dgr)):
        out[f'd_red/{n}'] = reductions(gr, SEED + 500 + i)
        if gr.numel() <= FULL_MAX:
            out[f'd_grad/{n}'] = gr.numpy()
    out['meta'] = np.array([S_, CAP, LAT, B, L, SEED])
    np.savez_compressed(os.path.join(HERE, 'nets_c3.npz'), **out)
    print('wrote nets_c3.npz with', len(out), 'arrays; logits', logits.detach().numpy().reshape(-1), 'gp', float(gp),
          '; G params', sum(p.numel() for p in G.parameters()), 'D params', sum(p.numel() for p in D.parameters()))


if __name__ == '__main__':
    main()
