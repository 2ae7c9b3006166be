"""`device='cpu'` of the drop-in histogram modules (histogan_amd/hist_cpu.py: PyTorch CPU ops, no HIP call) against the golden
vectors of the UNMODIFIED reference -- the same 1e-5 (forward) / 1e-4 (gradient, Hellinger gradient) / 1e-4 (loss) bars as
the GPU path.  Also the contract of SURVEY.md section 8b: works inside forked DataLoader workers, never initialises the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, golden_names, load_golden, relmax
from test_oracle_planes_golden import NAMES as PLANE_NAMES, load as load_plane


def _block(kwargs):
    from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
    kw = dict(kwargs)
    if kw.get('hist_boundary') is not None:
        kw['hist_boundary'] = list(kw['hist_boundary'])
    return RGBuvHistBlock(device='cpu', **kw)


@pytest.mark.parametrize('name', golden_names())
def test_cpu_module_matches_reference_golden(name):
    g = load_golden(name)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    out = _block(g['kwargs'])(x)
    assert out.device.type == 'cpu' and out.dtype == torch.float32 and out.shape == g['hist'].shape
    assert relmax(out.detach().numpy(), g['hist']) <= 1e-5
    out.backward(torch.from_numpy(g['grad_out']))
    assert relmax(x.grad.numpy(), g['grad_x']) <= 1e-4


@pytest.mark.parametrize('name', [n for n in golden_names() if 'hell_loss' in load_golden(n)])
def test_cpu_module_hellinger_matches_reference_golden(name):
    """The notebook's CPU path (configs[0]): histogram + Hellinger loss + gradient, Histogram_loss.ipynb:394-417."""
    g = load_golden(name)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    out = _block(g['kwargs'])(x)
    t = torch.from_numpy(g['target_hist'])
    loss = (1 / np.sqrt(2.0)) * torch.sqrt(torch.sum((torch.sqrt(t) - torch.sqrt(out)) ** 2)) / out.shape[0]
    loss.backward()
    assert abs(float(loss.detach()) - float(g['hell_loss'])) <= 1e-4
    assert relmax(x.grad.numpy(), g['hell_grad_x']) <= 1e-4


@pytest.mark.parametrize('name', ['c1_4x128', 'trainer_2x256to150', 'thr_1x256', 'jpeg1024to150'])
def test_cpu_module_matches_big_golden(name):
    """configs[0] (C1, 4 x 128^2: the notebook's CPU case) and the 256^2 / 1024^2 -> 150 cases, reference outputs;
    pre_relu=True is the train step's `histBlock(F.relu(x))` (histoGAN/histoGAN.py:955)."""
    import torch.nn.functional as F
    from bigcases import load_big
    g = load_big(name)
    spec = g['spec']
    x = g['x'].clone().requires_grad_(True)
    blk = _block(spec['kw'])
    out = blk(x, pre_relu=True) if spec.get('relu') else blk(x)
    assert relmax(out.detach().numpy(), g['hist']) <= 1e-5
    if spec.get('relu'):
        assert torch.equal(out.detach(), blk(F.relu(x.detach())))
    if 'hell_loss' in g:
        t = torch.from_numpy(g['target_hist'])
        loss = (1 / np.sqrt(2.0)) * torch.sqrt(torch.sum((torch.sqrt(t) - torch.sqrt(out)) ** 2)) / out.shape[0]
        (gx,) = torch.autograd.grad(loss, x)
        assert abs(float(loss.detach()) - float(g['hell_loss'])) <= 1e-4
        if 'hell_grad_x' in g:
            assert relmax(gx.numpy(), g['hell_grad_x']) <= 1e-4


@pytest.mark.parametrize('name', PLANE_NAMES)
def test_cpu_plane_modules_match_reference_golden(name):
    from histogram_classes.LabHistBlock import LabHistBlock
    from histogram_classes.rgChromaHistBlock import rgChromaHistBlock
    g = load_plane(name)
    kw = dict(g['kwargs'])
    if 'hist_boundary' in kw:
        kw['hist_boundary'] = list(kw['hist_boundary'])
    blk = (rgChromaHistBlock if g['projection'] == 'rgchroma' else LabHistBlock)(device='cpu', **kw)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    out = blk(x)
    assert relmax(out.detach().numpy(), g['hist']) <= 1e-5
    out.backward(torch.from_numpy(g['grad_out']))
    assert relmax(x.grad.numpy(), g['grad_x']) <= 1e-4


class _RefStyleDataset(torch.utils.data.Dataset):
    """What the reference's Dataset does (histoGAN/histoGAN.py:263-266, 296-302): the block is built with device='cpu'
    in the constructor and called per item inside the worker."""

    def __init__(self, n):
        from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
        self.block = RGBuvHistBlock(insz=150, h=16, resizing='sampling', method='inverse-quadratic', sigma=0.02, device='cpu')
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        img = torch.rand(3, 40, 56, generator=torch.Generator().manual_seed(i))
        with torch.no_grad():
            hist = self.block(img.unsqueeze(0)).squeeze(0)
        return {'images': img, 'histograms': hist, 'cuda_initialised': torch.cuda.is_initialized()}


def test_cpu_module_runs_inside_forked_dataloader_workers():
    ds = _RefStyleDataset(6)
    direct = [ds[i]['histograms'] for i in range(6)]
    loader = torch.utils.data.DataLoader(ds, batch_size=2, num_workers=2, shuffle=False)
    got = [b for b in loader]
    assert len(got) == 3
    for k, b in enumerate(got):
        assert b['histograms'].shape == (2, 3, 16, 16)
        assert not bool(b['cuda_initialised'].any()), 'device="cpu" touched the GPU inside a worker'
        for j in range(2):
            # (workers run single-threaded: the bmm's summation order differs from the main process's)
            assert relmax(b['histograms'][j].numpy(), direct[2 * k + j].numpy()) <= 1e-6
    assert abs(float(got[0]['histograms'][0].sum()) - 1.0) < 1e-5


def test_cpu_path_is_not_a_fallback_of_the_gpu_path():
    """CPU tensors never reach the HIP functions and GPU modules never take the CPU path: the two are selected by the
    module's `device` argument only; the autograd Function of the GPU path keeps refusing CPU tensors."""
    from histogan_amd.hist import HistConfig, RGBuvHistFunction
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        RGBuvHistFunction.apply(torch.rand(1, 3, 8, 8), HistConfig(h=16), False)
    import histogan_amd.hist_cpu as HC
    src = open(HC.__file__).read()
    assert 'oracle' not in src.replace('oracle/', '').split('"""', 2)[2], 'the product CPU path must not import the oracle'
    assert '_lib' not in src.split('"""', 2)[2]
