"""Host-side glue added in round 3 that is plain torch (runs on the CPU): the chunked skinny GEMM, the two-copy batch
concatenation and the batched mapping-network call.  The CUDA-only branches are covered by the -m gpu network tests."""
import pytest
import torch

from histogan_amd import ops
from histogan_amd import trainer as T


@pytest.mark.parametrize('B,R,C', [(32, 2048, 1024), (2, 1024, 2048), (4, 512, 64), (3, 300, 7), (5, 12288, 96), (1, 8192, 1)])
@pytest.mark.parametrize('transposed', [True, False])
def test_skinny_mm_equals_mm(B, R, C, transposed):
    g = torch.Generator().manual_seed(B + R + C)
    a = torch.randn(B, R, generator=g, dtype=torch.float64)
    m = torch.randn(C, R, generator=g, dtype=torch.float64) if transposed else torch.randn(R, C, generator=g, dtype=torch.float64)
    ref = a @ (m.t() if transposed else m)
    out = ops._skinny_mm(a, m, transposed)
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) <= 1e-11 * max(1.0, float(ref.abs().max()))
    # gradients flow through the chunked form like through mm
    a2, m2 = a.clone().requires_grad_(True), m.clone().requires_grad_(True)
    ops._skinny_mm(a2, m2, transposed).square().sum().backward()
    a3, m3 = a.clone().requires_grad_(True), m.clone().requires_grad_(True)
    (a3 @ (m3.t() if transposed else m3)).square().sum().backward()
    assert torch.allclose(a2.grad, a3.grad, rtol=1e-10, atol=1e-10) and torch.allclose(m2.grad, m3.grad, rtol=1e-10, atol=1e-10)


def test_skinny_mm_falls_back(monkeypatch):
    a, m = torch.randn(4, 2048), torch.randn(2048, 8)
    calls = []
    real_bmm = torch.bmm
    monkeypatch.setattr(torch, 'bmm', lambda *x: calls.append(1) or real_bmm(*x))
    ops._skinny_mm(a, m, False)
    assert calls                                   # chunked
    calls.clear()
    ops._skinny_mm(a[:, ::2], m[::2], False)       # non-contiguous operands: plain mm
    ops._skinny_mm(torch.randn(4, 300), torch.randn(300, 8), False)    # reduction too short / not divisible
    monkeypatch.setattr(ops, 'SKINNY_SPLIT', False)
    ops._skinny_mm(a, m, False)
    assert not calls


def test_cat_batches_equals_cat():
    a, b = torch.randn(3, 3, 8, 8), torch.randn(5, 3, 8, 8)
    assert torch.equal(T._cat_batches(a, b), torch.cat((a, b), 0))
    br = b.clone().requires_grad_(True)
    out = T._cat_batches(a, br)                    # a graph is needed: falls back to torch.cat
    assert out.requires_grad and torch.equal(out.detach(), torch.cat((a, b), 0))
    assert torch.equal(T._cat_batches(a, b.double()).double(), torch.cat((a.double(), b.double()), 0))   # dtype mismatch: cat's rules


def test_latent_to_w_layer_counts_and_fallback():
    S = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.LeakyReLU(0.2))
    z1, z2 = torch.randn(4, 16), torch.randn(4, 16)
    out = T.latent_to_w(S, [(z1, 3), (z2, 2)])
    assert [n for _, n in out] == [3, 2]
    assert torch.equal(out[0][0], S(z1)) and torch.equal(out[1][0], S(z2))
    t = T.styles_def_to_tensor(out)
    assert t.shape == (4, 5, 16) and torch.equal(t[:, 2], out[0][0]) and torch.equal(t[:, 3], out[1][0])


def test_demod_backward_formulas_equal_autograd_of_the_reference_expression():
    """The closed forms hg_demod_weight_term / hg_demod_style_grad implement (and the GPU tests compare the kernels with) are
    the gradients of the reference's demodulation, histoGAN/histoGAN.py:427-429, written there on per-sample weights:
        weights = w[None] * (y[:, None, :, None, None] + 1);  d = rsqrt((weights ** 2).sum(dim=(2, 3, 4)) + EPS)
    Checked here against autograd of exactly that expression in fp64."""
    g = torch.Generator().manual_seed(5)
    B, N, K, k = 3, 6, 5, 3
    w = torch.randn(N, K, k, k, generator=g, dtype=torch.float64, requires_grad=True)
    y = torch.randn(B, K, generator=g, dtype=torch.float64, requires_grad=True)
    gd = torch.randn(B, N, generator=g, dtype=torch.float64)
    weights = w[None] * (y[:, None, :, None, None] + 1)
    d = torch.rsqrt((weights ** 2).sum(dim=(2, 3, 4)) + 1e-8)
    gy_ref, gw_ref = torch.autograd.grad(d, (y, w), gd)
    with torch.no_grad():
        s1 = y + 1
        wsq = w.pow(2).sum(dim=(2, 3))
        assert torch.allclose(torch.rsqrt((s1 * s1) @ wsq.t() + 1e-8), d, rtol=1e-12, atol=0)      # the shared-weight form of d
        gq = gd * (-0.5) * d ** 3
        gy = 2.0 * s1 * (gq @ wsq)                                   # hg_demod_style_grad
        gw = 2.0 * w * (gq.t() @ (s1 * s1))[:, :, None, None]        # hg_demod_weight_term
    assert torch.allclose(gy, gy_ref, rtol=1e-10, atol=1e-12) and torch.allclose(gw, gw_ref, rtol=1e-10, atol=1e-12)
