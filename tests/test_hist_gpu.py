"""GPU parity: the HIP RGB-uv histogram (through the C ABI) vs golden vectors of the reference
and vs the oracle.  Tolerances: forward max|d|/max|ref| <= 1e-5,
gradient <= 1e-4, Hellinger loss |d| <= 1e-4."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, relmax

pytestmark = pytest.mark.gpu

FWD_TOL, BWD_TOL, LOSS_TOL = 1e-5, 1e-4, 1e-4


def _block(kwargs):
    from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
    kw = dict(kwargs)
    if 'hist_boundary' in kw:
        kw['hist_boundary'] = list(kw['hist_boundary'])
    return RGBuvHistBlock(device='cuda', **kw)


@pytest.mark.parametrize('name', golden_names())
def test_forward_matches_reference_golden(name, gpu_device):
    g = load_golden(name)
    out = _block(g['kwargs'])(torch.from_numpy(g['x']).to(gpu_device))
    assert out.shape == g['hist'].shape and out.dtype == torch.float32
    assert relmax(out.cpu().numpy(), g['hist']) <= FWD_TOL


@pytest.mark.parametrize('name', golden_names())
def test_backward_matches_reference_golden(name, gpu_device):
    g = load_golden(name)
    x = torch.from_numpy(g['x']).to(gpu_device).requires_grad_(True)
    out = _block(g['kwargs'])(x)
    out.backward(torch.from_numpy(g['grad_out']).to(gpu_device))
    assert relmax(x.grad.cpu().numpy(), g['grad_x']) <= BWD_TOL


@pytest.mark.parametrize('name', [n for n in golden_names() if 'hell_loss' in load_golden(n)])
def test_hellinger_matches_reference_golden(name, gpu_device):
    from histogan_amd.hist import hellinger_loss
    g = load_golden(name)
    x = torch.from_numpy(g['x']).to(gpu_device).requires_grad_(True)
    out = _block(g['kwargs'])(x)
    loss = hellinger_loss(torch.from_numpy(g['target_hist']).to(gpu_device), out)
    loss.backward()
    assert abs(float(loss) - float(g['hell_loss'])) <= LOSS_TOL
    assert relmax(x.grad.cpu().numpy(), g['hell_grad_x']) <= BWD_TOL


def test_hellinger_inline_formula_matches_kernel(gpu_device):
    """The reference's inline formula (histoGAN.py:957-960) on torch-ROCm vs the fused kernel."""
    from histogan_amd.hist import hellinger_loss
    torch.manual_seed(0)
    t = torch.rand(4, 3, 64, 64, device=gpu_device); t = t / t.sum(dim=(1, 2, 3), keepdim=True)
    gen = torch.rand(4, 3, 64, 64, device=gpu_device); gen = (gen / gen.sum(dim=(1, 2, 3), keepdim=True)).requires_grad_(True)
    ref = 2.0 * (1 / np.sqrt(2.0)) * torch.sqrt(torch.sum(torch.pow(torch.sqrt(t) - torch.sqrt(gen), 2))) / 4
    (gref,) = torch.autograd.grad(ref, gen)
    ours = hellinger_loss(t, gen, alpha=2.0)
    (gours,) = torch.autograd.grad(ours, gen)
    assert abs(float(ours) - float(ref)) <= 1e-6
    assert relmax(gours.cpu().numpy(), gref.cpu().numpy()) <= 1e-5


THR_CASES = [
    dict(h=16, insz=64, intensity_scale=True),                                   # eps < bin spacing: one bin per pixel
    dict(h=16, insz=64, intensity_scale=True, hist_boundary=[0.5, 3.0]),         # eps > spacing: two bins can hit
    dict(h=33, insz=40, intensity_scale=False, resizing='interpolation'),        # counts only, odd h, resize
    dict(h=128, insz=150, intensity_scale=True, green_only=True),                # 128 KB LDS grid
    dict(h=8, insz=20, intensity_scale=True, resizing='sampling'),
]


@pytest.mark.parametrize('kw', THR_CASES)
def test_thresholding_scatter_matches_oracle(kw, gpu_device):
    """method='thresholding' runs on the scatter-add kernels (k_hist_thr_fwd / _bwd): parity with the oracle's dense
    formulation, incl. boundaries where a value falls into two windows, and bit-identical repeat runs."""
    from oracle import rgbuv_hist as O
    g = torch.Generator().manual_seed(kw['h'])
    x = (torch.rand(3, 3, 48, 56, generator=g) * 1.2 - 0.1)                      # some values outside [0, 1]
    x[0, :, :4] = 0.0; x[1, :, 5:9, 5:9] = 1.0; x[2, 0] = 0.25                   # exact zeros / ones / constant plane
    blk = _block(dict(method='thresholding', sigma=0.02, **kw))
    xg = x.to(gpu_device).requires_grad_(True)
    out = blk(xg)
    go = torch.randn(out.shape, generator=g)
    out.backward(go.to(gpu_device))
    xo = x.clone().requires_grad_(True)
    ref = O.rgbuv_hist(xo, method='thresholding', **kw)
    assert out.shape == ref.shape
    assert relmax(out.detach().cpu().numpy(), ref.detach().numpy()) <= FWD_TOL
    if ref.requires_grad:
        ref.backward(go)
        assert relmax(xg.grad.cpu().numpy(), xo.grad.numpy()) <= BWD_TOL
    else:                                # no intensity scale: the window has no slope, the reference has no gradient path
        assert float(xg.grad.abs().max()) == 0.0
    again = blk(x.to(gpu_device))
    assert torch.equal(again, out.detach())                                      # integer LDS atomics: order-independent


def test_module_surface_matches_reference_contract(gpu_device):
    """ctor kwargs / attributes, device variants, lazy bare Exceptions with the reference's
    messages (RGBuvHistBlock.py:90-93, 141-144), empty state_dict, strided / double / C = 4 inputs."""
    from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
    from oracle import rgbuv_hist as O
    bnd = [3, -3]
    m = RGBuvHistBlock(h=16, insz=32, resizing='sampling', method='thresholding', hist_boundary=bnd, device='cuda')
    assert bnd == [-3, 3] and m.hist_boundary is bnd                      # sorted IN PLACE like the reference (:66-69)
    assert m.eps == 6 / 16 and not hasattr(m, 'sigma') and len(m.state_dict()) == 0 and len(list(m.parameters())) == 0
    m2 = RGBuvHistBlock(h=16, sigma=0.05)
    assert (m2.h, m2.insz, m2.resizing, m2.method, m2.sigma, m2.intensity_scale, m2.green_only) == \
        (16, 150, 'interpolation', 'inverse-quadratic', 0.05, True, False)
    x = torch.rand(2, 4, 20, 28)                                           # C = 4: first three channels are used
    ref = O.rgbuv_hist(x, h=16).numpy()
    for dev_arg in ('cuda', 0, torch.device('cuda:0'), gpu_device):
        out = RGBuvHistBlock(h=16, device=dev_arg)(x.to(gpu_device))
        assert out.is_cuda and out.dtype == torch.float32 and relmax(out.cpu().numpy(), ref) <= FWD_TOL
    assert relmax(RGBuvHistBlock(h=16)(x).cpu().numpy(), ref) <= FWD_TOL    # CPU tensor is moved to the module's GPU
    assert relmax(RGBuvHistBlock(h=16)(x.double().to(gpu_device)).cpu().numpy(), ref) <= FWD_TOL
    big = torch.rand(2, 4, 40, 56, device=gpu_device)
    view = big[:, :, ::2, ::2]                                             # non-contiguous view
    assert relmax(RGBuvHistBlock(h=16)(view).cpu().numpy(), O.rgbuv_hist(view.cpu(), h=16).numpy()) <= FWD_TOL
    # bad arguments are accepted by the constructor and raise a bare Exception in forward, reference messages
    with pytest.raises(Exception, match='Wrong kernel method'):
        RGBuvHistBlock(h=16, method='gaussian')(x.to(gpu_device))
    with pytest.raises(Exception, match='Wrong resizing method'):
        RGBuvHistBlock(h=16, insz=8, resizing='nearest')(x.to(gpu_device))
    RGBuvHistBlock(h=16, insz=64, resizing='nearest')(x.to(gpu_device))    # not reached when no resize is needed (:80-93)
    out_cpu = RGBuvHistBlock(h=16, device='cpu')(x)                         # device='cpu': the HIP-free CPU implementation
    assert out_cpu.device.type == 'cpu' and relmax(out_cpu.numpy(), ref) <= FWD_TOL
    assert RGBuvHistBlock(h=16, device='cpu')(x.to(gpu_device)).device.type == 'cpu'
    # sums to one over all planes; one plane when green_only
    s = RGBuvHistBlock(h=16)(x.to(gpu_device)).sum(dim=(1, 2, 3))
    assert torch.allclose(s, torch.ones_like(s), atol=1e-5)
    assert RGBuvHistBlock(h=16, green_only=True)(x.to(gpu_device)).shape == (2, 1, 16, 16)


RBF_CASES = [
    dict(h=64, insz=64, sigma=0.02),                                             # the default sigma: support radius 1 bin
    dict(h=16, insz=64, sigma=0.1, hist_boundary=[-2.0, 3.0]),                   # radius 2, asymmetric boundary
    dict(h=32, insz=40, sigma=0.03, resizing='interpolation', intensity_scale=False),
    dict(h=128, insz=150, sigma=0.02, green_only=True),                          # one 128 KB grid
    dict(h=24, insz=20, sigma=0.05, resizing='sampling'),
]


@pytest.mark.parametrize('kw', RBF_CASES)
def test_rbf_truncated_scatter_matches_oracle(kw, gpu_device, monkeypatch):
    """method='RBF' with a kernel narrower than ~2 bins runs on the truncated scatter / gather kernels (weights beyond
    the support are < 1e-12): parity with the oracle's dense formulation and with the dense MFMA path."""
    from oracle import rgbuv_hist as O
    g = torch.Generator().manual_seed(kw['h'] + 1)
    x = (torch.rand(2, 3, 48, 56, generator=g) * 1.2 - 0.1)
    x[0, :, :4] = 0.0; x[1, :, 5:9, 5:9] = 1.0
    blk = _block(dict(method='RBF', **kw))
    xg = x.to(gpu_device).requires_grad_(True)
    out = blk(xg)
    go = torch.randn(out.shape, generator=g)
    out.backward(go.to(gpu_device))
    xo = x.clone().requires_grad_(True)
    ref = O.rgbuv_hist(xo, method='RBF', **kw)
    ref.backward(go)
    assert relmax(out.detach().cpu().numpy(), ref.detach().numpy()) <= FWD_TOL
    # exact zeros of the histogram make the reference's own gradient NaN-free here (upstream gradient is random, not
    # the Hellinger one); compare where the reference is finite
    gr = xo.grad.numpy()
    assert np.isfinite(gr).all()
    assert relmax(xg.grad.cpu().numpy(), gr) <= BWD_TOL
    assert torch.equal(blk(x.to(gpu_device)), out.detach())                      # deterministic
    monkeypatch.setenv('HG_RBF_DENSE', '1')                                      # A/B: the dense MFMA formulation
    xd = x.to(gpu_device).requires_grad_(True)
    dense = blk(xd)
    dense.backward(go.to(gpu_device))
    assert relmax(out.detach().cpu().numpy(), dense.detach().cpu().numpy()) <= 2e-6
    assert relmax(xg.grad.cpu().numpy(), xd.grad.cpu().numpy()) <= 2e-5


PLANES_CASES = [
    dict(h=128, insz=150),                                                       # configs[4]: RT = 4, 66 KB of LDS
    dict(h=128, insz=64, hist_boundary=[-2.5, 3.5], sigma=0.05),                 # h = 128 and asymmetric
    dict(h=96, insz=40, resizing='interpolation'),                               # RT = 3, zero-padded rows, resize adjoint
    dict(h=64, insz=64, hist_boundary=[-2.0, 4.0]),                              # asymmetric at the default size
    dict(h=40, insz=64, hist_boundary=[-3.0, 1.0], intensity_scale=False),
    dict(h=16, insz=20, hist_boundary=[0.5, 3.0], resizing='sampling', green_only=True),
    dict(h=72, insz=64, method='RBF', sigma=0.5),                                # wide RBF: dense path
    dict(h=32, insz=64, method='RBF', sigma=0.4, hist_boundary=[-1.0, 3.0]),
]


@pytest.mark.parametrize('kw', PLANES_CASES)
def test_plane_backward_matches_oracle(kw, gpu_device):
    """k_hist_bwd_planes (asymmetric boundary / 64 < h <= 128 / one plane): gradient vs the oracle's autograd."""
    from oracle import rgbuv_hist as O
    g = torch.Generator().manual_seed(kw['h'] + 7)
    x = (torch.rand(2, 4, 48, 56, generator=g) * 1.2 - 0.1)
    x[0, :, :4] = 0.0; x[1, :, 5:9, 5:9] = 1.0
    blk = _block(kw)
    xg = x.to(gpu_device).requires_grad_(True)
    out = blk(xg)
    go = torch.randn(out.shape, generator=g)
    out.backward(go.to(gpu_device))
    xo = x.clone().requires_grad_(True)
    ref = O.rgbuv_hist(xo, **kw)
    ref.backward(go)
    assert relmax(out.detach().cpu().numpy(), ref.detach().numpy()) <= FWD_TOL
    assert relmax(xg.grad.cpu().numpy(), xo.grad.numpy()) <= BWD_TOL
    assert float(xg.grad[:, 3].abs().max()) == 0.0                              # channels >= 3 get zeros
    xg.grad = None
    blk(xg).backward(go.to(gpu_device))
    # bit-identical repeat (no atomics) -- except through the sampling adjoint's atomicAdd when a side < h
    if kw.get('resizing') != 'sampling':
        x2 = x.to(gpu_device).requires_grad_(True)
        blk(x2).backward(go.to(gpu_device))
        assert torch.equal(x2.grad, xg.grad)


@pytest.mark.parametrize('h,method,sigma', [(64, 'inverse-quadratic', 0.02), (32, 'inverse-quadratic', 0.05),
                                            (64, 'RBF', 0.3)])
def test_plane_backward_matches_merged_backward(h, method, sigma, gpu_device, monkeypatch):
    """Two MFMA formulations of the same gradient: the mirrored-table kernel (k_hist_bwd, default for a symmetric
    boundary and h <= 64) and the plane-at-a-time kernel (HG_BWD_PLANES=1)."""
    g = torch.Generator().manual_seed(h)
    x = torch.rand(3, 3, 64, 80, generator=g)
    blk = _block(dict(h=h, insz=128, method=method, sigma=sigma))
    go = (torch.rand(3, 3, h, h, generator=g) - 0.3).to(gpu_device)
    xa = x.to(gpu_device).requires_grad_(True)
    blk(xa).backward(go)
    monkeypatch.setenv('HG_BWD_PLANES', '1')
    xb = x.to(gpu_device).requires_grad_(True)
    blk(xb).backward(go)
    assert relmax(xb.grad.cpu().numpy(), xa.grad.cpu().numpy()) <= 1e-5


def test_fast_window_classification_is_exact(gpu_device, monkeypatch):
    """method='thresholding': the scatter kernels decide window membership with the hardware logarithm + an explicit
    error margin and fall back to the fp64 path inside the margin.  (i) the logarithm's error bound the margins assume,
    measured exhaustively over the input range; (ii) histograms and gradients bit-identical to the all-exact path
    (HG_THR_EXACT=1) on 2.1 M pixels incl. saturated / dark / constant regions."""
    import ctypes
    from histogan_amd._lib import check, lib
    out2 = torch.zeros(2, device=gpu_device)
    check(lib.hg_selftest_fastlog(out2.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'selftest')
    rel, ab = out2.cpu().tolist()
    print(f'fast log: max rel err {rel:.3e} (|ln x| >= 1e-3), max abs err {ab:.3e} (near x = 1)')
    assert rel <= 3.0e-7 and ab <= 3.0e-7          # margins: 3.6e-7 relative, 9e-7 absolute (incl. the bin arithmetic's roundings)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(32, 3, 256, 256, generator=g)
    x[0] = torch.rand(3, 256, 256, generator=g) * 0.01               # dark image: |ln x| large
    x[1] = 1.0 - torch.rand(3, 256, 256, generator=g) * 1e-3         # near-saturated: ln x ~ 0
    x[2, :, :128] = 0.0; x[2, :, 128:] = 1.0
    x[3] = torch.rand(3, 1, 1, generator=g)                          # constant colour
    x[4] = (torch.rand(3, 256, 256, generator=g) * 255).round() / 255  # 8-bit image values
    for kw in (dict(h=64, insz=256), dict(h=64, insz=150), dict(h=32, insz=256, green_only=True)):
        blk = _block(dict(method='thresholding', **kw))
        go = None
        res = []
        for exact in ('0', '1'):
            monkeypatch.setenv('HG_THR_EXACT', exact)
            xg = x.to(gpu_device).requires_grad_(True)
            out = blk(xg)
            if go is None:
                go = torch.randn(out.shape, generator=g).to(gpu_device)
            out.backward(go)
            res.append((out.detach().clone(), xg.grad.clone()))
        assert torch.equal(res[0][0], res[1][0])
        assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize('kw', [dict(h=32, insz=64), dict(h=16, insz=24, resizing='interpolation'),
                                dict(h=16, insz=64, resizing='sampling'), dict(h=32, insz=64, method='thresholding'),
                                dict(h=16, insz=64, hist_boundary=[-2.0, 3.0])])
def test_pre_relu_equals_relu_in_front(kw, gpu_device):
    """forward(x, pre_relu=True) == forward(F.relu(x)) in value and gradient (the train step's call, histoGAN.py:955):
    generator-like input with negative values, exact zeros (the one point where relu's mask differs from clamp's)
    and values above 1."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    x = 0.5 + 0.7 * torch.randn(2, 3, 40, 48, generator=g)
    x[0, :, :6] = 0.0
    x[1, 1, 10:20] = 0.0
    blk = _block(kw)
    go = None
    res = []
    for fused in (False, True):
        xg = x.to(gpu_device).requires_grad_(True)
        out = blk(xg, pre_relu=True) if fused else blk(F.relu(xg))
        if go is None:
            go = (torch.rand(out.shape, generator=g) - 0.3).to(gpu_device)
        out.backward(go)
        res.append((out.detach(), xg.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    assert float(res[1][1][0, :, :6].abs().max()) == 0.0           # relu's mask at x == 0
