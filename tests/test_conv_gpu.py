"""GPU parity of the fp32-MFMA implicit-GEMM convolutions (include/hg_conv.h) against torch's fp64
convolution of the same op (output, data gradient, weight gradient, bias gradient), through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import relmax

pytestmark = pytest.mark.gpu

# (B, K, N, H, W, ksize)
CASES = [
    (2, 5, 7, 4, 4, 3), (3, 16, 33, 8, 8, 3), (1, 3, 16, 37, 53, 3), (2, 64, 32, 64, 64, 3),
    (2, 8, 3, 32, 32, 1), (4, 130, 70, 16, 16, 3), (2, 32, 32, 2, 2, 3), (1, 1, 1, 1, 1, 3),
    (3, 40, 200, 4, 4, 3), (2, 24, 64, 128, 128, 3), (2, 64, 3, 64, 64, 1), (5, 17, 19, 8, 8, 1),
    (2, 3, 16, 32, 32, 1), (8, 96, 160, 8, 8, 3), (1, 20, 48, 16, 40, 3), (33, 9, 6, 4, 4, 3),
]


@pytest.mark.parametrize('B,K,N,H,W,k', CASES)
def test_conv2d_same_matches_fp64(B, K, N, H, W, k, gpu_device):
    from histogan_amd.conv import conv2d_same
    g = torch.Generator(device='cpu').manual_seed(B * 1000 + K * 10 + N + H)
    x = torch.randn(B, K, H, W, generator=g).to(gpu_device).requires_grad_(True)
    w = (torch.randn(N, K, k, k, generator=g) / (K * k * k) ** 0.5).to(gpu_device).requires_grad_(True)
    b = torch.randn(N, generator=g).to(gpu_device).requires_grad_(True)
    go = torch.randn(B, N, H, W, generator=g).to(gpu_device)

    out = conv2d_same(x, w, b)
    gx, gw, gb = torch.autograd.grad(out, (x, w, b), go)

    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = F.conv2d(xd, wd, bd, padding=k // 2)
    rx, rw, rb = torch.autograd.grad(ref, (xd, wd, bd), go.double())

    assert out.shape == ref.shape
    assert relmax(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 2e-6
    assert relmax(gx.cpu().numpy(), rx.cpu().numpy()) <= 2e-6
    assert relmax(gw.cpu().numpy(), rw.cpu().numpy()) <= 5e-6
    assert relmax(gb.cpu().numpy(), rb.cpu().numpy()) <= 5e-6


def test_conv_fused_scales(gpu_device):
    """iscale (modulation) / oscale (demodulation) / bias fused into the kernel == the unfused expression."""
    from histogan_amd import conv as C
    torch.manual_seed(3)
    B, K, N, H, W = 3, 24, 40, 16, 16
    x = torch.randn(B, K, H, W, device=gpu_device)
    w = torch.randn(N, K, 3, 3, device=gpu_device) / (K * 9) ** 0.5
    s = torch.rand(B, K, device=gpu_device) + 0.5
    d = torch.rand(B, N, device=gpu_device) + 0.5
    bias = torch.randn(N, device=gpu_device)
    out = C.conv_packed(x, C.pack_weights(w, C.PACK_FWD), N, 3, iscale=s, oscale=d, bias=bias)
    ref = F.conv2d((x * s[:, :, None, None]).double(), w.double(), padding=1) * d[:, :, None, None].double() \
        + bias[None, :, None, None].double()
    assert relmax(out.cpu().numpy(), ref.cpu().numpy()) <= 2e-6
    go = torch.randn(B, N, H, W, device=gpu_device)
    gw = C.conv_wgrad(x, go, 3, iscale=s, gscale=d)
    xd = (x * s[:, :, None, None]).double()
    wd = w.double().requires_grad_(True)
    rw, = torch.autograd.grad(F.conv2d(xd, wd, padding=1), wd, (go * d[:, :, None, None]).double())
    assert relmax(gw.cpu().numpy(), rw.cpu().numpy()) <= 5e-6


def test_conv_is_deterministic(gpu_device):
    from histogan_amd.conv import conv2d_same
    torch.manual_seed(4)
    x = torch.randn(4, 32, 32, 32, device=gpu_device, requires_grad=True)
    w = torch.randn(32, 32, 3, 3, device=gpu_device, requires_grad=True)
    go = torch.randn(4, 32, 32, 32, device=gpu_device)
    a = torch.autograd.grad(conv2d_same(x, w), (x, w), go)
    b = torch.autograd.grad(conv2d_same(x, w), (x, w), go)
    assert all(torch.equal(p, q) for p, q in zip(a, b))


def test_conv_rejects_cpu_and_bad_kernel(gpu_device):
    from histogan_amd.conv import conv2d_same
    with pytest.raises(RuntimeError):
        conv2d_same(torch.randn(1, 2, 4, 4), torch.randn(3, 2, 3, 3))
    with pytest.raises(ValueError):
        conv2d_same(torch.randn(1, 2, 8, 8, device=gpu_device), torch.randn(3, 2, 5, 5, device=gpu_device))
