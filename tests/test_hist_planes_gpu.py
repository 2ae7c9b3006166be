"""GPU parity of the rg-chroma / Lab histogram blocks (projection variants of the hist kernels, through the C ABI)
against golden vectors of the reference's rgChromaHistBlock / LabHistBlock.  Same bars as the RGB-uv block:
forward max|d|/max|ref| <= 1e-5, gradient <= 1e-4."""
import pytest
import torch

from conftest import relmax
from test_oracle_planes_golden import NAMES, load

pytestmark = pytest.mark.gpu


def _block(g):
    from histogram_classes.LabHistBlock import LabHistBlock
    from histogram_classes.rgChromaHistBlock import rgChromaHistBlock
    kw = dict(g['kwargs'])
    if 'hist_boundary' in kw:
        kw['hist_boundary'] = list(kw['hist_boundary'])
    return (rgChromaHistBlock if g['projection'] == 'rgchroma' else LabHistBlock)(device='cuda', **kw)


@pytest.mark.parametrize('name', NAMES)
def test_plane_hist_forward_backward(name, gpu_device):
    g = load(name)
    x = torch.from_numpy(g['x']).to(gpu_device).requires_grad_(True)
    out = _block(g)(x)
    assert out.shape == g['hist'].shape and out.dtype == torch.float32
    assert relmax(out.detach().cpu().numpy(), g['hist']) <= 1e-5
    out.backward(torch.from_numpy(g['grad_out']).to(gpu_device))
    assert relmax(x.grad.cpu().numpy(), g['grad_x']) <= 1e-4
