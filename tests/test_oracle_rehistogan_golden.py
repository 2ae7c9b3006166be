"""Pin oracle/rehistogan_nets.py to golden vectors produced by the reference's own ReHistoGAN classes."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, relmax
from oracle import histogan_nets as N
from oracle import rehistogan_nets as RN

VARIANTS = dict(plain=(False, False), skip=(True, False), skipint=(True, True))


@pytest.fixture(scope='module')
def g():
    z = np.load(os.path.join(GOLDEN_DIR, 'rehistogan_small.npz'))
    return {k: z[k] for k in z.files}


def sd_of(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix + '/')}


def run_variant(g, tag, mod=RN):
    skip, internal = VARIANTS[tag]
    S = int(g['meta'][0])
    ed = {k: v.requires_grad_(True) for k, v in sd_of(g, f'{tag}/ED').items()}
    gg = {k: v.requires_grad_(True) for k, v in sd_of(g, f'{tag}/G').items()}
    x = torch.from_numpy(g['img']).requires_grad_(True)
    hw = torch.from_numpy(g['hw'])
    res = mod.encoder_decoder(ed, x, hw if internal else torch.from_numpy(g['hist']), S, skip, internal)
    gen = mod.recoloring_head(gg, res[0], hw, torch.from_numpy(g['noise']), *(res[2:] if skip else ()))
    return ed, gg, x, res, gen


@pytest.mark.parametrize('tag', list(VARIANTS))
def test_encoder_decoder_and_head(g, tag):
    ed, gg, x, res, gen = run_variant(g, tag)
    assert relmax(res[0].detach().numpy(), g[f'{tag}/latent']) <= 2e-6
    assert relmax(res[1].detach().numpy(), g[f'{tag}/rgb']) <= 2e-6
    if VARIANTS[tag][0]:
        assert relmax(res[2].detach().numpy(), g[f'{tag}/p1']) <= 2e-6
        assert relmax(res[3].detach().numpy(), g[f'{tag}/p2']) <= 2e-6
    assert relmax(gen.detach().numpy(), g[f'{tag}/gen']) <= 2e-6
    en = [k[len(f'{tag}/ed_grad/'):] for k in g if k.startswith(f'{tag}/ed_grad/')]
    gn = [k[len(f'{tag}/g_grad/'):] for k in g if k.startswith(f'{tag}/g_grad/')]
    grads = torch.autograd.grad(gen, [x] + [ed[n] for n in en] + [gg[n] for n in gn], torch.from_numpy(g[f'{tag}/go']))
    assert relmax(grads[0].numpy(), g[f'{tag}/gx']) <= 2e-5
    for n, gr in zip(en, grads[1:1 + len(en)]):
        assert relmax(gr.numpy(), g[f'{tag}/ed_grad/{n}']) <= 2e-5, n
    for n, gr in zip(gn, grads[1 + len(en):]):
        assert relmax(gr.numpy(), g[f'{tag}/g_grad/{n}']) <= 2e-5, n


def test_hist_vectorizer(g):
    assert relmax(N.vectorizer(sd_of(g, 'H'), torch.from_numpy(g['hist']), 'fcs').numpy(), g['hw']) <= 1e-6


@pytest.mark.parametrize('kind,tag', [('L1', 'l1'), ('1st gradient', 'sobel'), ('2nd gradient', 'lap')])
def test_reconstruction_loss(g, kind, tag):
    a = torch.from_numpy(g['loss_a'])
    b = torch.from_numpy(g['loss_b']).requires_grad_(True)
    v = RN.rec_loss(kind, a, b)
    assert abs(float(v) - float(g[f'rec_{tag}'])) <= 1e-6 * max(1.0, abs(float(g[f'rec_{tag}'])))
    gr, = torch.autograd.grad(v, b)
    assert relmax(gr.numpy(), g[f'rec_{tag}_grad']) <= 1e-5


def test_gaussian_and_variance_loss(g):
    k = RN.gaussian_kernel(15, 5, 3)
    assert np.array_equal(k.numpy(), g['gauss_k'])
    a = torch.from_numpy(g['loss_a'])
    b = torch.from_numpy(g['loss_b']).requires_grad_(True)
    assert relmax(RN.gaussian_op(a, k).numpy(), g['gauss_out']) <= 1e-6
    v = RN.variance_loss(1.5, torch.from_numpy(g['hist']), torch.from_numpy(g['hist2']), a, b, k)
    assert abs(float(v) - float(g['var_loss'])) <= 1e-6 * max(1.0, abs(float(g['var_loss'])))
    gr, = torch.autograd.grad(v, b)
    assert relmax(gr.numpy(), g['var_grad']) <= 1e-5


def test_histogram_loss(g):
    v = RN.histogram_loss(32, torch.from_numpy(g['hist']), torch.from_numpy(g['hist2']))
    assert abs(float(v) - float(g['hist_loss'])) <= 1e-6 * abs(float(g['hist_loss']))
