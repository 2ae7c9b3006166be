"""Pin oracle.rgbuv_hist.plane_hist (rg-chroma / Lab one-plane histograms) to golden vectors of the unmodified
reference classes (tests/golden/make_golden_planes.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, relmax
from oracle import rgbuv_hist as O

with open(os.path.join(GOLDEN_DIR, 'PLANES_INDEX.json')) as f:
    NAMES = json.load(f)


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, f'plane_{name}.npz'))
    rec = {k: z[k] for k in z.files}
    rec['kwargs'] = json.loads(str(rec['kwargs']))
    rec['projection'] = str(rec['projection'])
    return rec


@pytest.mark.parametrize('name', NAMES)
def test_oracle_plane_hist_matches_reference(name):
    g = load(name)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    out = O.plane_hist(x, g['projection'], **g['kwargs'])
    assert out.shape == g['hist'].shape
    assert relmax(out.detach().numpy(), g['hist']) <= 1e-6
    if True:
        (gx,) = torch.autograd.grad(out, x, torch.from_numpy(g['grad_out']))
        assert relmax(gx.numpy(), g['grad_x']) <= 1e-5
