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
