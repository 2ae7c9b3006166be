"""DiffAugment kernels (include/hg_augment.h) against the outputs of the reference's
utils/diff_augment.py (recorded draws replayed), their adjoints, and the augmented train step."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, relmax
from test_oracle_augment_golden import cases, steps_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def g():
    z = np.load(os.path.join(GOLDEN_DIR, 'augment.npz'))
    return {k: z[k] for k in z.files}


def test_replay_matches_reference(g, gpu_device):
    from histogan_amd.augment import augment_color, augment_spatial
    for name in cases(g):
        x = torch.from_numpy(g[f'{name}/x']).to(gpu_device).requires_grad_(True)
        y = x
        for kind, rows in steps_of(g, name):
            y = augment_spatial(y, rows) if kind == 'spatial' else augment_color(y, rows)
        col = 'color' in name
        assert relmax(y.detach().cpu().numpy(), g[f'{name}/y']) <= (2e-6 if col else 0.0), name
        go = torch.from_numpy(g[f'{name}/go']).to(gpu_device).requires_grad_(True)
        gx, = torch.autograd.grad(y, x, go, create_graph=True)
        assert relmax(gx.detach().cpu().numpy(), g[f'{name}/gx']) <= (1e-5 if col else 0.0), name
        # linear chain: d <gx, r> / d go == forward(r) - forward(0)   (backward of the backward is the forward map)
        r = torch.randn_like(x)
        g2, = torch.autograd.grad((gx * r).sum(), go)
        def f(t):
            for kind, rows in steps_of(g, name):
                t = augment_spatial(t, rows) if kind == 'spatial' else augment_color(t, rows)
            return t
        ref2 = f(r) - f(torch.zeros_like(r))
        assert relmax(g2.cpu().numpy(), ref2.cpu().numpy()) <= 1e-5, name


def test_fused_run_equals_two_launches(g, gpu_device):
    """flip + translation + cutout in ONE parameter row == the reference's sequence of three ops."""
    from histogan_amd.augment import augment_spatial
    name = 'translation+cutout'
    x = torch.from_numpy(g[f'{name}/x']).to(gpu_device)
    (k0, r0), (k1, r1) = steps_of(g, name)
    rows = r0.copy()
    rows[:, 5:9] = r1[:, 5:9]
    assert np.array_equal(augment_spatial(x, rows).cpu().numpy(), g[f'{name}/y'])
    # the flip is applied FIRST: flipping the input by hand and running the row without the flip bit is the same
    noflip = rows.copy()
    rows[:, 0] = 1
    assert torch.equal(augment_spatial(x, rows), augment_spatial(torch.flip(x, dims=(3,)), noflip))


def test_flip_and_random_draws(g, gpu_device):
    from histogan_amd.augment import AugWrapper, DiffAugment
    x = torch.from_numpy(g['flip/x']).to(gpu_device)
    assert np.array_equal(DiffAugment(x, [], flip=1).cpu().numpy(), g['flip/y'])
    big = torch.rand(16, 3, 64, 64, device=gpu_device) + 1.0
    gen = torch.Generator().manual_seed(4)
    a = DiffAugment(big, ['translation', 'cutout'], generator=gen)
    b = DiffAugment(big, ['translation', 'cutout'], generator=torch.Generator().manual_seed(4))
    assert torch.equal(a, b) and a.shape == big.shape
    zero_frac = float((a == 0).float().mean())
    assert 0.15 < zero_frac < 0.45          # cutout removes <= 25 %, translation up to ~23 % more
    c = DiffAugment(big, ['color', 'offset'], generator=gen)
    assert torch.isfinite(c).all() and c.shape == big.shape
    w = AugWrapper(torch.nn.Identity())
    assert torch.equal(w(big, prob=0.0, types=['cutout']), big)
    assert not torch.equal(w(big, prob=1.0, types=['cutout']), big)
    with pytest.raises(KeyError):
        DiffAugment(big, ['nope'])
    with pytest.raises(RuntimeError):
        DiffAugment(big.cpu(), ['cutout'])


def test_train_steps_with_augmentation(gpu_device, tmp_path):
    from histoGAN import Trainer
    tr = Trainer('aug', tmp_path / 'r', tmp_path / 'm', 32, 2, batch_size=2, hist_bin=16, hist_insz=32,
                 hist_resizing='interpolation', aug_prob=1.0, aug_types=['translation', 'cutout', 'color'])
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    for _ in range(2):                       # step 0: gradient penalty through the augmented real images
        tr.train(alpha=2)
    assert tr.GAN.D_aug is not None and np.isfinite(tr.d_loss) and np.isfinite(tr.g_loss) and np.isfinite(tr.last_gp_loss)
    assert any(k.startswith('D_aug.D.') for k in tr.GAN.state_dict())      # the reference's checkpoint layout


def test_non_default_ratios(gpu_device):
    """utils.diff_augment with the ratio arguments of the reference's functions (diff_augment.py:33-97): the draws obey
    the ratio's bounds and the result is the gather the drawn parameters describe."""
    from utils import diff_augment as U
    x = torch.rand(6, 3, 20, 24, device=gpu_device)
    torch.manual_seed(1)
    y = U.rand_translation(x, ratio=0.25)                # shifts within +-5 rows / +-6 columns, zero fill
    assert y.shape == x.shape
    found = 0
    for b in range(6):
        ok = False
        for sh in range(-5, 6):
            for sw in range(-6, 7):
                ref = torch.zeros_like(x[b])
                ys0, ys1 = max(0, -sh), min(20, 20 - sh)
                xs0, xs1 = max(0, -sw), min(24, 24 - sw)
                ref[:, ys0:ys1, xs0:xs1] = x[b][:, ys0 + sh:ys1 + sh, xs0 + sw:xs1 + sw]
                if torch.equal(ref, y[b]):
                    ok = True
        found += int(ok)
    assert found == 6
    z = U.rand_cutout(x, ratio=0.25)                     # a 5 x 6 hole (clipped at the border) of zeros per sample
    for b in range(6):
        hole = (z[b] == 0).all(dim=0)
        assert (z[b][:, ~hole] == x[b][:, ~hole]).all()
        rows, cols = hole.any(dim=1).sum().item(), hole.any(dim=0).sum().item()
        assert 1 <= rows <= 5 and 1 <= cols <= 6
    w = U.rand_offset(x, ratio=0.5, ratio_h=1, ratio_v=0)   # a roll along W only, |roll| <= int(H*0.5)
    for b in range(6):
        assert any(torch.equal(torch.roll(x[b], r, dims=2), w[b]) for r in range(-10, 11))
