"""Winograd F(2x2, 3x3) convolutions (include/hg_wino.h) through the C ABI against torch's fp64 convolution: the output, the
data gradient (transposed / flipped operand), the weight gradient, the fused epilogue of the generator stage (modulation,
demodulation, noise, LeakyReLU: histoGAN/histoGAN.py:420-440, 465-476), the residual addend of the discriminator block
(:520-524) -- on ragged maps (tile masks), batch tails inside an image group, both channel-block variants, forced K splits,
launches with more tiles than CUs (the persistent tile loop), the padding corners of the 16-byte row loads; determinism; and
the autograd dispatch (conv2d takes the Winograd form exactly where hg_wino_supported says so, with the same gradients).

Bars: 2e-6 max-norm relative for outputs and data gradients, 4e-6 for weight gradients -- tighter than the direct kernel's
5e-6 / 1e-5 (tests/test_c3_parity_gpu.py): sixteen accumulation chains of depth K instead of one of depth 9 K."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300))


def _p(t):
    return None if t is None else t.data_ptr()


def _pack(w, mode):
    from histogan_amd._lib import check, lib, raw_stream
    Co, Ci = w.shape[:2]
    n = lib.hg_wino_packed_elems(Co, Ci, mode)
    assert n > 0
    u = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.hg_wino_pack_weights(w.data_ptr(), u.data_ptr(), Co, Ci, mode, raw_stream(w.device)), 'hg_wino_pack_weights')
    return u


def _conv(x, u, N, **kw):
    from histogan_amd import conv as C
    return C.wino_conv(x, u, N, **kw)


def _wgrad(x, go):
    from histogan_amd._lib import check, lib, raw_stream
    B, K, H, W = x.shape
    N = go.shape[1]
    nb = lib.hg_wino_wgrad_workspace_bytes(B, K, N, H, W)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    gw = torch.empty((N, K, 3, 3), dtype=torch.float32, device=x.device)
    check(lib.hg_wino_wgrad(x.data_ptr(), go.data_ptr(), gw.data_ptr(), B, K, N, H, W, ws.data_ptr(), nb, raw_stream(x.device)),
          'hg_wino_wgrad')
    return gw


CASES = [(3, 32, 64, 8, 8), (2, 64, 128, 16, 16), (5, 40, 96, 12, 20), (1, 64, 64, 6, 10), (7, 32, 32, 8, 8),
         (2, 16, 32, 32, 32), (3, 64, 32, 24, 16), (2, 128, 192, 4, 4), (33, 64, 64, 2, 2), (2, 72, 80, 34, 30),
         (3, 72, 80, 16, 32), (5, 64, 128, 8, 4), (2, 128, 64, 64, 64), (9, 8, 32, 64, 64), (4, 64, 96, 128, 128)]


@pytest.mark.parametrize('B,K,N,H,W', CASES, ids=lambda v: str(v))
def test_wino_output_dgrad_epilogues_match_fp64(B, K, N, H, W, gpu_device):
    from histogan_amd._lib import lib
    dev = gpu_device
    g = torch.Generator().manual_seed(B * 131 + K + N + H)
    x = torch.randn(B, K, H, W, generator=g).to(dev)
    w = (torch.randn(N, K, 3, 3, generator=g) / (K * 9) ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    go = torch.randn(B, N, H, W, generator=g).to(dev)
    assert lib.hg_wino_packed_elems(N, K, 0)
    uf = _pack(w, 0)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    out = _conv(x, uf, N, bias=bias)
    assert out.shape == ref.shape and _rel(out, ref) <= 2e-6
    assert torch.equal(out, _conv(x, uf, N, bias=bias))              # deterministic
    if lib.hg_wino_packed_elems(N, K, 1):                             # data gradient: the launch's K is the conv's N
        refd = torch.nn.grad.conv2d_input((B, K, H, W), w.double(), go.double(), padding=1)
        assert _rel(_conv(go, _pack(w, 1), K), refd) <= 2e-6
    # fused generator-stage epilogue
    isc = (torch.randn(B, K, generator=g) * 0.3 + 1).to(dev)
    osc = (torch.rand(B, N, generator=g) + 0.5).to(dev)
    S = max(H, W) + (max(H, W) & 1) + 2
    nimg = torch.randn(B, S, S, generator=g).to(dev)
    nw = torch.randn(N, generator=g).to(dev)
    reff = F.conv2d(x.double() * isc.double()[:, :, None, None], w.double(), padding=1) * osc.double()[:, :, None, None] \
        + bias.double()[None, :, None, None] + nw.double()[None, :, None, None] * nimg.double()[:, None, :H, :W]
    reff = F.leaky_relu(reff, 0.2)
    outf = _conv(x, uf, N, iscale=isc, oscale=osc, bias=bias, noise_w=nw, noise_img=nimg, noise_S=S, slope=0.2)
    assert _rel(outf, reff) <= 2e-6
    ad = torch.randn(B, N, H, W, generator=g).to(dev)
    assert _rel(_conv(x, uf, N, bias=bias, addend=ad), ref + ad.double()) <= 2e-6


@pytest.mark.parametrize('B,K,N,H,W', [c for c in CASES if c[3] & (c[3] - 1) == 0 and c[4] & (c[4] - 1) == 0] +
                         [(32, 64, 64, 16, 16), (6, 200, 72, 8, 16)], ids=lambda v: str(v))
def test_wino_wgrad_matches_fp64(B, K, N, H, W, gpu_device):
    dev = gpu_device
    g = torch.Generator().manual_seed(B * 17 + K + 3 * N + H)
    x = torch.randn(B, K, H, W, generator=g).to(dev)
    go = torch.randn(B, N, H, W, generator=g).to(dev)
    ref = torch.nn.grad.conv2d_weight(x.double(), (N, K, 3, 3), go.double(), padding=1)
    gw = _wgrad(x, go)
    assert gw.shape == ref.shape and _rel(gw, ref) <= 4e-6
    assert torch.equal(gw, _wgrad(x, go))                             # deterministic (fixed-order slab sums)


def test_wino_forced_ksplit_and_variant_in_subprocess(gpu_device):
    """HG_WINO_KSPLIT (K-split slabs + k_wino_reduce, incl. the fused epilogue in the reduce) and HG_WINO_WG_SPLITS are read
    once per process: run them in children."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import test_wino_gpu as T
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
B, K, N, H, W = 3, 128, 96, 12, 8
x = torch.randn(B, K, H, W, generator=g).to(dev); w = (torch.randn(N, K, 3, 3, generator=g) / (K * 9) ** 0.5).to(dev)
bias = torch.randn(N, generator=g).to(dev); osc = (torch.rand(B, N, generator=g) + 0.5).to(dev)
nimg = torch.randn(B, 12, 12, generator=g).to(dev); nw = torch.randn(N, generator=g).to(dev)
ref = F.leaky_relu(F.conv2d(x.double(), w.double(), padding=1) * osc.double()[:, :, None, None] + bias.double()[None, :, None, None]
                   + nw.double()[None, :, None, None] * nimg.double()[:, None, :H, :W], 0.2)
out = T._conv(x, T._pack(w, 0), N, oscale=osc, bias=bias, noise_w=nw, noise_img=nimg, noise_S=12, slope=0.2)
assert T._rel(out, ref) <= 2e-6, T._rel(out, ref)
x2 = torch.randn(4, 64, 16, 16, generator=g).to(dev); go = torch.randn(4, 64, 16, 16, generator=g).to(dev)
refw = torch.nn.grad.conv2d_weight(x2.double(), (64, 64, 3, 3), go.double(), padding=1)
assert T._rel(T._wgrad(x2, go), refw) <= 4e-6
print("ok")
''' % (root, root)
    for env in ({'HG_WINO_KSPLIT': '4', 'HG_WINO_WG_SPLITS': '5'}, {'HG_WINO_KSPLIT': '2', 'HG_WINO_PERSIST': '0', 'HG_WINO_WG_SPLITS': '40'}):
        r = subprocess.run([sys.executable, '-c', code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and 'ok' in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize('B,K,N,S', [(32, 256, 128, 64), (64, 64, 64, 64), (32, 64, 32, 256), (8, 2048, 1024, 8)], ids=lambda v: str(v))
def test_conv2d_dispatches_winograd_with_same_gradients(B, K, N, S, gpu_device):
    """histogan_amd.conv.conv2d (autograd) at shapes the library serves on the Winograd form: output, data, weight and bias
    gradients against fp64, and the SAME call with the dispatch switched off (direct kernels) within the sum of both bars."""
    from histogan_amd import conv as C
    dev = gpu_device
    assert C.wino_supported(B, K, N, S, S)
    g = torch.Generator().manual_seed(K + N + S)
    nb = min(B, 8)      # (fp64 reference on a batch slice: samples are independent; the weight gradient on the slice alone)
    x = torch.randn(B, K, S, S, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(N, K, 3, 3, generator=g) / (K * 9) ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(B, N, S, S, generator=g).to(dev)
    out = C.conv2d(x, w, b, 1)
    gx, gw, gb = torch.autograd.grad(out, (x, w, b), go)
    xs = x[:nb].detach().double().requires_grad_(True)
    wd, bd = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    ref = F.conv2d(xs, wd, bd, padding=1)
    rx, = torch.autograd.grad(ref, (xs,), go[:nb].double())
    assert _rel(out[:nb].detach(), ref.detach()) <= 2e-6 and _rel(gx[:nb], rx) <= 2e-6
    outs = C.conv2d(x[:nb].detach().requires_grad_(True), w, b, 1)
    gws, gbs = torch.autograd.grad(outs, (w, b), go[:nb])
    rw, rb = torch.autograd.grad(F.conv2d(xs, wd, bd, padding=1), (wd, bd), go[:nb].double())
    assert _rel(gws, rw) <= 1e-5 and _rel(gbs, rb) <= 1e-5
    # against the direct kernels on the full batch
    saved = C._wino_u, C.wino_wgrad_supported
    try:
        C._wino_u = lambda *a, **k: None
        C.wino_wgrad_supported = lambda *a: False
        outd = C.conv2d(x, w, b, 1)
        gxd, gwd, gbd = torch.autograd.grad(outd, (x, w, b), go)
    finally:
        C._wino_u, C.wino_wgrad_supported = saved
    assert _rel(out.detach(), outd.detach().double()) <= 7e-6 and _rel(gx, gxd.double()) <= 7e-6 and _rel(gw, gwd.double()) <= 1.4e-5
    assert torch.equal(gb, gbd)


def test_wino_supported_is_host_logic(gpu_device):
    from histogan_amd._lib import lib
    assert lib.hg_wino_supported(32, 256, 128, 64, 64) == 1 and lib.hg_wino_wgrad_supported(32, 256, 128, 64, 64) == 1
    assert lib.hg_wino_supported(32, 256, 128, 63, 64) == 0            # odd map
    assert lib.hg_wino_supported(32, 3, 16, 256, 256) == 0             # 3 input channels
    assert lib.hg_wino_supported(1, 64, 64, 4, 4) == 0                 # too few tiles to fill the chip
    assert lib.hg_wino_wgrad_supported(32, 64, 32, 256, 256) == 0      # < 64 channels on a side
    assert lib.hg_wino_wgrad_workspace_bytes(32, 64, 64, 24, 24) == 0  # tiles per side not a power of two
    assert lib.hg_wino_packed_elems(128, 36, 0) == 0 and lib.hg_wino_packed_elems(128, 40, 0) > 0
