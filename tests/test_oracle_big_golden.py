"""Pin the oracle at the BASELINE shapes: oracle/rgbuv_hist.py vs outputs of the unmodified reference on the
configs[0]/[1]-shaped inputs (tests/golden/make_golden_big.py).  A subset (the whole set costs ~1 min of CPU)."""
import pytest
import torch
import torch.nn.functional as F

from bigcases import load_big
from conftest import relmax
from oracle import rgbuv_hist as O


@pytest.mark.parametrize('name', ['c1_4x128', 'trainer_2x256to150', 'thr_1x256'])
def test_oracle_matches_reference_at_baseline_shapes(name):
    g = load_big(name)
    spec = g['spec']
    x = g['x'].clone().requires_grad_(True)
    out = O.rgbuv_hist(F.relu(x) if spec.get('relu') else x, **spec['kw'])
    assert relmax(out.detach().numpy(), g['hist']) <= 1e-6
    if 'hell_loss' in g:
        loss = O.hellinger_loss(torch.from_numpy(g['target_hist']), out)
        (gx,) = torch.autograd.grad(loss, x)
        assert abs(float(loss) - float(g['hell_loss'])) <= 1e-6
        assert relmax(gx.numpy(), g['hell_grad_x']) <= 1e-5
