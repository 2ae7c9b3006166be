"""Pin oracle/diff_augment.py to outputs of the reference's utils/diff_augment.py (recorded random draws replayed)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, relmax
from oracle import diff_augment as OA


@pytest.fixture(scope='module')
def g():
    z = np.load(os.path.join(GOLDEN_DIR, 'augment.npz'))
    return {k: z[k] for k in z.files}


def cases(g):
    return sorted({k.split('/')[0] for k in g if k.endswith('/gx')})


def steps_of(g, name):
    ks = sorted((k for k in g if k.startswith(name + '/step')), key=lambda k: int(k.split('step')[1].split('_')[0]))
    return [(k.rsplit('_', 1)[1], g[k]) for k in ks]


def replay(g, name, x, spatial, color):
    for kind, rows in steps_of(g, name):
        x = spatial(x, rows) if kind == 'spatial' else color(x, rows)
    return x


def test_all_cases(g):
    assert len(cases(g)) == 10
    for name in cases(g):
        x = torch.from_numpy(g[f'{name}/x']).requires_grad_(True)
        y = replay(g, name, x, OA.spatial, OA.color)
        tol = 1e-6 if 'color' in name else 0.0
        assert relmax(y.detach().numpy(), g[f'{name}/y']) <= tol, name
        gx, = torch.autograd.grad(y, x, torch.from_numpy(g[f'{name}/go']))
        assert relmax(gx.numpy(), g[f'{name}/gx']) <= (1e-5 if 'color' in name else 0.0), name


def test_flip(g):
    x = torch.from_numpy(g['flip/x'])
    rows = [[1, 0, 0, 0, 0, 1, 0, 1, 0]] * x.shape[0]
    assert np.array_equal(OA.spatial(x, rows).numpy(), g['flip/y'])


def test_host_draws_have_the_reference_supports():
    """The per-sample parameter draws of histogan_amd/augment.py cover exactly the integer ranges of the reference's
    torch.randint / random.randint calls (utils/diff_augment.py:34-38, 58-62, 80-83)."""
    from histogan_amd import augment as A
    gen = torch.Generator().manual_seed(0)
    for H, W in ((16, 16), (15, 21), (32, 8)):
        sh, sw = A.draw_translation(4000, H, W, generator=gen)
        rh, rw = int(H * 0.125 + 0.5), int(W * 0.125 + 0.5)
        assert set(sh.tolist()) == set(range(-rh, rh + 1)) and set(sw.tolist()) == set(range(-rw, rw + 1))
        r0, r1, c0, c1 = A.draw_cutout(6000, H, W, generator=gen)
        ch, cw = int(H * 0.5 + 0.5), int(W * 0.5 + 0.5)
        offs_h = range(0, H + (1 - ch % 2))
        assert {(a, b) for a, b in zip(r0.tolist(), r1.tolist())} == {OA.cutout_box(o, 0, H, W)[:2] for o in offs_h}
        offs_w = range(0, W + (1 - cw % 2))
        assert {(a, b) for a, b in zip(c0.tolist(), c1.tolist())} == {OA.cutout_box(0, o, H, W)[2:] for o in offs_w}
        vh, vv = A.draw_offset(6000, H, W, generator=gen)
        assert set(vh.tolist()) == {2 * k - H for k in range(H + 1)} and set(vv.tolist()) == {2 * k - W for k in range(W + 1)}
        vh, vv = A.draw_offset(500, H, W, 1, 1, 0, generator=gen)        # offset_h: no vertical roll
        assert set(vv.tolist()) == {0}
