"""Pin oracle/histogan_nets.py to the reference AT THE BENCH WIDTH (256^2, network_capacity 16, latent 512): goldens from
the unmodified reference Generator / Discriminator / gradient_penalty (histoGAN/histoGAN.py:529-631, 156-163) run on the CPU
at B = 1 by tests/golden/make_golden_nets_c3.py.  The 83 M + 91 M weights are rebuilt from the seed (`synth_state_dict`,
guarded by a fingerprint); stored are rgb, logits, the penalty, every small gradient tensor in full and two fp64
reductions (signed sum, L2 norm) of EVERY parameter gradient.  tests/test_c3_parity_gpu.py checks the HIP networks against
the same file on the GPU."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, relmax
from oracle import histogan_nets as N


def _mk():
    spec = importlib.util.spec_from_file_location('make_golden_nets_c3', os.path.join(GOLDEN_DIR, 'make_golden_nets_c3.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope='module')
def c3():
    z = np.load(os.path.join(GOLDEN_DIR, 'nets_c3.npz'))
    g = {k: z[k] for k in z.files}
    mk = _mk()
    specs = json.loads(str(g['spec']))
    seed = int(g['meta'][5])
    sds = {}
    for i, tag in enumerate(('G', 'D')):
        sd = mk.synth_state_dict(specs[tag], seed + i)
        assert np.allclose(mk.fingerprint(sd), g[f'{tag}_fingerprint'], rtol=1e-9, atol=0), 'seeded weights differ from the golden run'
        sds[tag] = sd
    gen = torch.Generator(device='cpu').manual_seed(seed + 2)
    S_, CAP, LAT, B, L, _ = [int(v) for v in g['meta']]
    inputs = dict(styles=torch.randn(B, L - 2, LAT, generator=gen), hists=torch.randn(B, 2, LAT, generator=gen),
                  noise=torch.rand(B, S_, S_, 1, generator=gen), go=torch.randn(B, 3, S_, S_, generator=gen),
                  img=torch.rand(B, 3, S_, S_, generator=torch.Generator(device='cpu').manual_seed(int(g['img_seed']))))
    return g, mk, sds, inputs


def _check_grads(g, mk, prefix, names, grads, seed0, tol):
    seed = int(g['meta'][5])
    for i, (n, gr) in enumerate(zip(names, grads)):
        red = mk.reductions(gr, seed + seed0 + i)
        ref = g[f'{prefix}_red/{n}']
        # signed sum against the tensor's norm (the sum of +-g_i is ~ norm in size), norm relative
        assert abs(red[0] - ref[0]) <= tol * max(ref[1], 1e-30), (n, red, ref)
        assert abs(red[1] - ref[1]) <= tol * max(ref[1], 1e-30), (n, red, ref)
        if f'{prefix}_grad/{n}' in g:
            assert relmax(gr.numpy(), g[f'{prefix}_grad/{n}']) <= tol, n


def test_generator_c3_width_matches_reference(c3):
    g, mk, sds, inp = c3
    L = int(g['meta'][4])
    sd = {k: v.clone().requires_grad_(True) for k, v in sds['G'].items()}
    styles, hists = inp['styles'].clone().requires_grad_(True), inp['hists'].clone().requires_grad_(True)
    rgb = N.generator(sd, styles, hists, inp['noise'], L)
    assert relmax(rgb.detach().numpy(), g['g_rgb']) <= 1e-5
    names = list(sd.keys())
    grads = torch.autograd.grad(rgb, [styles, hists] + [sd[n] for n in names], inp['go'])
    assert relmax(grads[0].numpy(), g['g_grad_styles']) <= 1e-4
    assert relmax(grads[1].numpy(), g['g_grad_hists']) <= 1e-4
    _check_grads(g, mk, 'g', names, grads[2:], 100, 1e-4)


def test_discriminator_and_penalty_c3_width_match_reference(c3):
    g, mk, sds, inp = c3
    L = int(g['meta'][4])
    sd = {k: v.clone().requires_grad_(True) for k, v in sds['D'].items()}
    x = inp['img'].clone().requires_grad_(True)
    logits = N.discriminator(sd, x, L + 1)
    assert relmax(logits.detach().numpy().reshape(-1), g['d_logits']) <= 1e-5
    gp = N.gradient_penalty(x, logits.reshape(1))
    assert abs(float(gp) - float(g['d_gp'])) <= 1e-4 * max(1.0, abs(float(g['d_gp'])))
    loss = torch.relu(1 + logits).mean() + gp
    assert abs(float(loss) - float(g['d_loss'])) <= 1e-4 * max(1.0, abs(float(g['d_loss'])))
    names = list(sd.keys())
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    _check_grads(g, mk, 'd', names, grads, 500, 1e-4)
