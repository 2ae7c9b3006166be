"""Pin the oracle (oracle/rgbuv_hist.py) to golden vectors produced by the unmodified reference."""
import pytest
import torch

from conftest import golden_names, load_golden, relmax
from oracle import rgbuv_hist as O


@pytest.mark.parametrize('name', golden_names())
def test_oracle_forward_matches_reference(name):
    g = load_golden(name)
    out = O.rgbuv_hist(torch.from_numpy(g['x']), **g['kwargs']).numpy()
    assert out.shape == g['hist'].shape
    # same op order and dtypes as the reference => agreement to fp32 rounding
    assert relmax(out, g['hist']) <= 1e-6


@pytest.mark.parametrize('name', golden_names())
def test_oracle_backward_matches_reference(name):
    g = load_golden(name)
    _, gx = O.rgbuv_hist_fwd_bwd(torch.from_numpy(g['x']), grad_out=torch.from_numpy(g['grad_out']),
                                 **g['kwargs'])
    assert relmax(gx.numpy(), g['grad_x']) <= 1e-5


@pytest.mark.parametrize('name', [n for n in golden_names() if 'hell_loss' in load_golden(n)])
def test_oracle_hellinger_matches_reference(name):
    g = load_golden(name)
    _, gx, loss = O.rgbuv_hist_fwd_bwd(torch.from_numpy(g['x']), target=torch.from_numpy(g['target_hist']),
                                       **g['kwargs'])
    assert abs(float(loss) - float(g['hell_loss'])) <= 1e-6
    assert relmax(gx.numpy(), g['hell_grad_x']) <= 1e-5


def test_truth_mode_close_to_fp32_mode():
    g = load_golden('iq_h64_b2_48')
    x = torch.from_numpy(g['x'])
    a = O.rgbuv_hist(x, **g['kwargs']).numpy()
    t = O.rgbuv_hist(x, truth=True, **g['kwargs']).numpy()
    assert relmax(a, t) <= 1e-5
