"""Pin oracle/histogan_nets.py to golden vectors produced by the reference's own classes."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, relmax
from oracle import histogan_nets as N


@pytest.fixture(scope='module')
def g():
    z = np.load(os.path.join(GOLDEN_DIR, 'nets_small.npz'))
    return {k: z[k] for k in z.files}


def sd_of(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix + '/')}


@pytest.mark.parametrize('tag,demod', [('c3', True), ('c1', False)])
def test_conv2d_mod(g, tag, demod):
    x = torch.from_numpy(g[f'{tag}/x']).requires_grad_(True)
    y = torch.from_numpy(g[f'{tag}/y']).requires_grad_(True)
    w = torch.from_numpy(g[f'{tag}/weight']).requires_grad_(True)
    o = N.conv2d_mod(x, y, w, demod)
    assert relmax(o.detach().numpy(), g[f'{tag}/out']) <= 1e-6
    gx, gy, gw = torch.autograd.grad(o, (x, y, w), torch.from_numpy(g[f'{tag}/go']))
    for a, b in ((gx, 'gx'), (gy, 'gy'), (gw, 'gw')):
        assert relmax(a.numpy(), g[f'{tag}/{b}']) <= 1e-5


def test_vectorizers(g):
    assert relmax(N.vectorizer(sd_of(g, 'S'), torch.from_numpy(g['z']), 'net').numpy(), g['w']) <= 1e-6
    assert relmax(N.vectorizer(sd_of(g, 'H'), torch.from_numpy(g['hist']), 'fcs').numpy(), g['hw']) <= 1e-6


def test_generator_forward_backward(g):
    sd = {k: v.requires_grad_(True) for k, v in sd_of(g, 'G').items()}
    L = int(g['meta'][5])
    styles = torch.from_numpy(g['g_styles']).requires_grad_(True)
    hists = torch.from_numpy(g['g_hists']).requires_grad_(True)
    rgb = N.generator(sd, styles, hists, torch.from_numpy(g['g_noise']), L)
    assert relmax(rgb.detach().numpy(), g['g_rgb']) <= 1e-6
    names = [k[len('g_grad/'):] for k in g if k.startswith('g_grad/')]
    grads = torch.autograd.grad(rgb, [styles, hists] + [sd[n] for n in names], torch.from_numpy(g['g_go']))
    assert relmax(grads[0].numpy(), g['g_grad_styles']) <= 1e-5
    assert relmax(grads[1].numpy(), g['g_grad_hists']) <= 1e-5
    for n, gr in zip(names, grads[2:]):
        assert relmax(gr.numpy(), g[f'g_grad/{n}']) <= 1e-5, n


def test_discriminator_and_gradient_penalty(g):
    sd = {k: v.requires_grad_(True) for k, v in sd_of(g, 'D').items()}
    img = torch.from_numpy(g['d_img']).requires_grad_(True)
    logits = N.discriminator(sd, img, int(g['meta'][5]) + 1)
    assert relmax(logits.detach().numpy(), g['d_logits']) <= 1e-6
    gp = N.gradient_penalty(img, logits)
    assert abs(float(gp) - float(g['d_gp'])) <= 1e-5 * max(1.0, abs(float(g['d_gp'])))
    loss = torch.relu(1 + logits).mean() + gp
    names = [k[len('d_grad/'):] for k in g if k.startswith('d_grad/')]
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gr in zip(names, grads):
        assert relmax(gr.numpy(), g[f'd_grad/{n}']) <= 1e-5, n


def test_styles_def_to_tensor(g):
    L = int(g['meta'][5])
    t = N.styles_def_to_tensor([(torch.from_numpy(g['styles_def_a']), 2), (torch.from_numpy(g['styles_def_b']), L - 4)])
    assert np.array_equal(t.numpy(), g['styles_def'])
