"""GPU parity of the fp32-MFMA implicit-GEMM convolutions (include/hg_conv.h) against torch's fp64
convolution of the same op (output, data gradient, weight gradient, bias gradient), through the C ABI."""
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
    # <= 16 output channels on wide maps: the 16x16x4 MFMA tile (first discriminator block, to-RGB)
    (2, 3, 16, 64, 64, 3), (2, 16, 16, 64, 64, 3), (3, 16, 3, 32, 32, 3), (2, 40, 3, 32, 32, 1), (1, 16, 16, 20, 44, 3),
    (2, 7, 12, 16, 16, 3), (2, 16, 16, 8, 8, 3),
    # one pixel chunk, 4-wave tiles: the weight gradient is a single slab (LDS-transposed direct store; K % 4 != 0 falls back)
    (1, 64, 64, 8, 8, 3), (1, 68, 100, 8, 8, 3), (1, 66, 70, 8, 8, 3), (2, 128, 96, 4, 4, 3),
    # few pixels, many channels: the 128x128 tile with a K split (forward 132->500, data gradient 500->132 ... both > 64)
    (16, 132, 500, 8, 8, 3), (16, 256, 260, 8, 8, 3), (6, 192, 640, 12, 12, 3),
    # 4x4 maps, many channels: the small-map instantiation of the 128x128 tile (8 images per pixel tile) + K split
    (64, 132, 1000, 4, 4, 3), (40, 520, 136, 4, 4, 3),
    # block counts that select the per-launch variants on a 256-CU chip (hg_conv.hip dispatch_conv / pick_ksplit_128):
    # 1024 blocks of the 128x128 tile and of the 64x256 tile -> the 2-channel K-chunk kernels at 4 blocks per CU (forward;
    # the data gradients of these layers take the 16- / 32-channel tiles); 256 blocks with a deep K -> K split 3 (768 blocks)
    (8, 8, 128, 128, 128, 3), (4, 8, 64, 256, 256, 3), (8, 128, 128, 64, 64, 3),
    # 1024 blocks of the 1x1 128x128 tile (3 blocks per CU by registers = 1.33 rounds): LDS-padded to 2 per CU
    (8, 16, 128, 128, 128, 1),
]


# stride-2 cases (3x3): the discriminator's down-sampling convolution (histoGAN/histoGAN.py:517-518)
CASES_S2 = [(2, 16, 16, 64, 64), (2, 16, 16, 32, 32), (3, 5, 7, 9, 13), (2, 64, 64, 16, 16), (4, 70, 130, 8, 8), (2, 32, 32, 64, 64),
            (1, 3, 4, 5, 4), (8, 128, 128, 4, 4), (2, 16, 16, 128, 128), (5, 33, 65, 2, 2), (1, 2, 2, 1, 1),
            # >= 64 pixel tiles per parity class: the data gradient's one-launch form with the XCD-paired block order, odd
            # sizes (the four classes differ by a row / column), each of the large-map tile shapes (16 / 32 / 64 / 128 channels)
            (8, 16, 16, 255, 257), (6, 24, 32, 127, 130), (12, 48, 64, 97, 61), (20, 96, 128, 63, 65)]


@pytest.mark.parametrize('B,K,N,H,W', CASES_S2)
def test_conv2d_stride2_matches_fp64(B, K, N, H, W, gpu_device):
    from histogan_amd.conv import conv2d
    g = torch.Generator(device='cpu').manual_seed(B * 1000 + K * 10 + N + H)
    x = torch.randn(B, K, H, W, generator=g).to(gpu_device).requires_grad_(True)
    w = (torch.randn(N, K, 3, 3, generator=g) / (K * 9) ** 0.5).to(gpu_device).requires_grad_(True)
    b = torch.randn(N, generator=g).to(gpu_device).requires_grad_(True)
    out = conv2d(x, w, b, stride=2)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = F.conv2d(xd, wd, bd, stride=2, padding=1)
    assert out.shape == ref.shape
    go = torch.randn(ref.shape, generator=g).to(gpu_device)
    gx, gw, gb = torch.autograd.grad(out, (x, w, b), go)
    rx, rw, rb = torch.autograd.grad(ref, (xd, wd, bd), go.double())
    assert relmax(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 2e-6
    assert relmax(gx.cpu().numpy(), rx.cpu().numpy()) <= 2e-6
    assert relmax(gw.cpu().numpy(), rw.cpu().numpy()) <= 5e-6
    assert relmax(gb.cpu().numpy(), rb.cpu().numpy()) <= 5e-6


@pytest.mark.parametrize('stride,k', [(1, 3), (2, 3), (1, 1)])
def test_conv2d_double_backward(stride, k, gpu_device):
    """Gradient-penalty pattern (histoGAN/histoGAN.py:156-163): d/dw and d/dx of || d out / d x ||^2 through a
    conv -> lrelu -> conv stack, against torch's fp64 double backward."""
    from histogan_amd.conv import conv2d, input_grads_only
    torch.manual_seed(11)
    B, K, N, H = 3, 6, 10, 12
    x = torch.randn(B, K, H, H, device=gpu_device, requires_grad=True)
    w1 = (torch.randn(N, K, k, k, device=gpu_device) / (K * k * k) ** 0.5).requires_grad_(True)
    b1 = torch.randn(N, device=gpu_device, requires_grad=True)
    w2 = (torch.randn(4, N, 3, 3, device=gpu_device) / (N * 9) ** 0.5).requires_grad_(True)

    def penalty(conv, x, w1, b1, w2, ctx):
        h = F.leaky_relu(conv(x, w1, b1, stride), 0.2)
        out = conv(h, w2, None, 1).pow(2).sum(dim=(1, 2, 3))
        with ctx():
            gr, = torch.autograd.grad(out, x, torch.ones_like(out), create_graph=True)
        return ((gr.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()

    import contextlib
    ours = penalty(conv2d, x, w1, b1, w2, input_grads_only)
    g_ours = torch.autograd.grad(ours, (x, w1, b1, w2))
    xd, w1d, b1d, w2d = (t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2))
    ref = penalty(lambda a, w, b, s: F.conv2d(a, w, b, stride=s, padding=w.shape[2] // 2), xd, w1d, b1d, w2d,
                  contextlib.nullcontext)
    g_ref = torch.autograd.grad(ref, (xd, w1d, b1d, w2d))
    assert abs(float(ours) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    for a, b in zip(g_ours, g_ref):
        assert relmax(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5


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
    out = C.conv_fwd_packed(x, C.pack_weights(w, C.PACK_FWD), N, 3, iscale=s, oscale=d, bias=bias)
    ref = F.conv2d((x * s[:, :, None, None]).double(), w.double(), padding=1) * d[:, :, None, None].double() \
        + bias[None, :, None, None].double()
    assert relmax(out.cpu().numpy(), ref.cpu().numpy()) <= 2e-6
    go = torch.randn(B, N, H, W, device=gpu_device)
    gw = C.conv_wgrad(x, go, 3, iscale=s, gscale=d)
    # data gradient with the scales swapped: gx = s * dgrad(d * go)
    gx = C.conv_dgrad_packed(go, C.pack_weights(w, C.PACK_DGRAD), K, H, W, 3, iscale=d, oscale=s)
    xr = x.double().requires_grad_(True)
    rx, = torch.autograd.grad(F.conv2d(xr * s[:, :, None, None].double(), w.double(), padding=1), xr,
                              (go * d[:, :, None, None]).double())
    assert relmax(gx.cpu().numpy(), rx.cpu().numpy()) <= 2e-6
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
    from histogan_amd.conv import conv2d
    with pytest.raises(ValueError):
        conv2d(torch.randn(1, 2, 8, 8, device=gpu_device), torch.randn(3, 2, 1, 1, device=gpu_device), None, 2)


def test_conv2d_lrelu_and_its_double_backward(gpu_device):
    """conv + bias + LeakyReLU in one launch (DiscriminatorBlock.net pairs) incl. the gradient-penalty pattern."""
    from histogan_amd.conv import conv2d_lrelu, input_grads_only
    torch.manual_seed(12)
    B, K, N, H = 3, 6, 10, 12
    x = torch.randn(B, K, H, H, device=gpu_device, requires_grad=True)
    w1 = (torch.randn(N, K, 3, 3, device=gpu_device) / (K * 9) ** 0.5).requires_grad_(True)
    b1 = torch.randn(N, device=gpu_device, requires_grad=True)
    w2 = (torch.randn(4, N, 3, 3, device=gpu_device) / (N * 9) ** 0.5).requires_grad_(True)
    b2 = torch.randn(4, device=gpu_device, requires_grad=True)

    def penalty(f, x, w1, b1, w2, b2, ctx):
        out = f(f(x, w1, b1), w2, b2)
        val = out.pow(2).sum(dim=(1, 2, 3))
        with ctx():
            gr, = torch.autograd.grad(val, x, torch.ones_like(val), create_graph=True)
        return out, ((gr.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean() + val.mean()

    import contextlib
    out, ours = penalty(lambda a, w, b: conv2d_lrelu(a, w, b, 0.2), x, w1, b1, w2, b2, input_grads_only)
    g_ours = torch.autograd.grad(ours, (x, w1, b1, w2, b2))
    dd = [t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    ref_out, ref = penalty(lambda a, w, b: F.leaky_relu(F.conv2d(a, w, b, padding=1), 0.2), *dd, contextlib.nullcontext)
    g_ref = torch.autograd.grad(ref, dd)
    assert relmax(out.detach().cpu().numpy(), ref_out.detach().cpu().numpy()) <= 2e-6
    assert abs(float(ours) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    for a, b in zip(g_ours, g_ref):
        assert relmax(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5


def test_batched_packing_launch_matches_single_packs_and_leaves_wsq(gpu_device):
    """hg_conv_pack_weights_multi over a flat buffer of registered weights: both packed operands equal the per-weight
    hg_conv_pack_weights results bit for bit, and hg_pack_item.wsq holds sum_taps W^2 (the weight factor of the demodulation
    coefficient, histoGAN/histoGAN.py:427-429) for every weight -- odd channel counts, 1x1 and 3x3."""
    from histogan_amd import conv as C
    from histogan_amd.optim import FlatParams
    torch.manual_seed(9)
    ws = [torch.nn.Parameter(torch.randn(s, device=gpu_device)) for s in ((40, 24, 3, 3), (3, 70, 1, 1), (130, 33, 3, 3))]
    flat = FlatParams(ws, with_grad=False)
    C.enable_pack_cache(ws)
    try:
        for w in ws:
            wq = C.cached(w, 'wsq', lambda t: (_ for _ in ()).throw(AssertionError('wsq must come from the packing launch')))
            assert torch.allclose(wq, w.detach().pow(2).sum(dim=(2, 3)), rtol=2e-6, atol=1e-7)
            for mode in (C.PACK_FWD, C.PACK_DGRAD):
                wt = C.pack_weights(w, mode)
                # (a 3x3 weight with both Winograd operands is left out of the batched DIRECT pack until a launch asks for
                # its direct operand: _direct_operand packs it then -- and the batched Winograd pack equals the single one)
                if getattr(wt, 'wino', False) is not False:
                    assert torch.equal(wt.wino, C._wino_pack(w.detach(), mode))
                assert torch.equal(C._direct_operand(wt), C._pack_weights(w.detach(), mode))
        with torch.no_grad():
            flat.data.mul_(2.0)
        C.weights_changed(flat.data)
        wq = C.cached(ws[2], 'wsq', lambda t: None)
        assert torch.allclose(wq, ws[2].detach().pow(2).sum(dim=(2, 3)), rtol=2e-6, atol=1e-7)
    finally:
        C.enable_pack_cache(None)


@pytest.mark.parametrize('B,K,N,H,k,stride', [(2, 3, 16, 64, 1, 1), (4, 16, 32, 32, 1, 1), (8, 520, 300, 4, 1, 1),
                                              (64, 1024, 2048, 2, 1, 1), (3, 24, 40, 16, 3, 1), (2, 16, 16, 32, 3, 2)])
def test_conv2d_add_equals_conv_then_add(B, K, N, H, k, stride, gpu_device):
    """hg_conv2d_fwd_add (the residual sum of DiscriminatorBlock in the conv_res epilogue, incl. the K-split launches of the
    small maps): bit-identical to conv2d(...) + addend, same gradients (the addend's is the incoming one), and the second
    order works (gradient penalty)."""
    from histogan_amd.conv import conv2d, conv2d_add
    torch.manual_seed(B + K + N)
    x = torch.randn(B, K, H, H, device=gpu_device, requires_grad=True)
    w = (torch.randn(N, K, k, k, device=gpu_device) / (K * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(N, device=gpu_device, requires_grad=True)
    Ho = (H - 1) // stride + 1
    ad = torch.randn(B, N, Ho, Ho, device=gpu_device, requires_grad=True)
    ref = conv2d(x, w, b, stride) + ad
    out = conv2d_add(x, w, b, ad, stride)
    assert torch.equal(out, ref)
    go = torch.randn_like(out)
    for a, r in zip(torch.autograd.grad(out, (x, w, b, ad), go), torch.autograd.grad(ref, (x, w, b, ad), go)):
        assert torch.equal(a, r)
    # second order: d/dw of || d out / d x ||^2
    def pen(fn):
        o = fn().pow(2).sum()
        gx, = torch.autograd.grad(o, x, create_graph=True)
        return torch.autograd.grad(gx.pow(2).sum(), (w, ad))
    for a, r in zip(pen(lambda: conv2d_add(x, w, b, ad, stride)), pen(lambda: conv2d(x, w, b, stride) + ad)):
        assert relmax(a.cpu().numpy(), r.cpu().numpy()) <= 1e-6


@pytest.mark.parametrize('shape', [(3, 5, 4, 4), (2, 16, 64, 64), (7, 3, 5, 9), (1, 1, 1, 1), (4, 130, 8, 8)])
def test_lrelu_bwd_channel_sum_matches_aten(shape, gpu_device):
    from histogan_amd.conv import lrelu_bwd_channel_sum
    torch.manual_seed(6)
    g, out = torch.randn(*shape, device=gpu_device), torch.randn(*shape, device=gpu_device)
    out[0, 0, 0, 0] = 0.0                                                 # the boundary: slope side (result > 0 is false)
    ref = torch.ops.aten.leaky_relu_backward(g, out, 0.2, True)
    gm, cs = lrelu_bwd_channel_sum(g, out, 0.2)
    assert torch.equal(gm, ref)
    assert relmax(cs.cpu().numpy(), ref.double().sum(dim=(0, 2, 3)).cpu().numpy()) <= 1e-6
    gm2, none = lrelu_bwd_channel_sum(g, out, 0.2, want_sum=False)
    assert none is None and torch.equal(gm2, ref)


@pytest.mark.parametrize('B,K,N,H', [(32, 2048, 1024, 8), (16, 2048, 2048, 4), (64, 1024, 2048, 2), (32, 1024, 512, 16)])
def test_splitk_launches_are_deterministic_and_exact(B, K, N, H, gpu_device):
    """K-split launches (few pixels, many channels: the plan says K split > 1) sum their slabs in fixed z order
    (k_splitk_reduce): results repeat bit for bit and match fp64; a forward and a data gradient."""
    import ctypes
    from histogan_amd._lib import lib
    from histogan_amd.conv import conv2d_same
    plan = (ctypes.c_int32 * 5)()
    assert lib.hg_conv2d_plan(B, K, N, H, H, 3, 1, 0, plan) == 0 and plan[1] > 1
    torch.manual_seed(K + N + H)
    x = torch.randn(B, K, H, H, device=gpu_device, requires_grad=True)
    w = (torch.randn(N, K, 3, 3, device=gpu_device) / (K * 9) ** 0.5).requires_grad_(True)
    go = torch.randn(B, N, H, H, device=gpu_device)
    outs = []
    for _ in range(6):
        y = conv2d_same(x, w)
        gx, = torch.autograd.grad(y, x, go)
        outs.append((y.detach(), gx))
    assert all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
    ref = F.conv2d(x.detach().double(), w.detach().double(), padding=1)
    assert relmax(outs[0][0].cpu().numpy(), ref.cpu().numpy()) <= 5e-6


@pytest.mark.parametrize('B,N,K,k', [(32, 64, 48, 3), (4, 5, 3, 3), (3, 7, 6, 1), (32, 1024, 2048, 3), (2, 33, 17, 1)])
def test_demod_weight_term_kernel_matches_fp64(B, N, K, k, gpu_device):
    """hg_demod_weight_term (the demodulation coefficient's weight gradient, written / accumulated into the flat gradient
    slot) against the aten formula of ops._DemodCoeff.backward in fp64 (reference: autograd through
    histoGAN/histoGAN.py:427-429)."""
    from histogan_amd._lib import check, lib, raw_stream
    g = torch.Generator(device='cpu').manual_seed(B + 7 * N + 13 * K + k)
    w = (torch.randn(N, K, k, k, generator=g) / (K * k * k) ** 0.5).to(gpu_device)
    s1 = (torch.randn(B, K, generator=g) * 0.5 + 1.0).to(gpu_device)
    gd = torch.randn(B, N, generator=g).to(gpu_device)
    d = torch.rsqrt(((s1 * s1) @ w.pow(2).sum(dim=(2, 3)).t()) + 1e-8)
    wd, sd, gdd, dd = (t.double() for t in (w, s1, gd, d))
    ref = 2.0 * wd * ((gdd * (-0.5) * dd ** 3).t() @ (sd * sd))[:, :, None, None]
    prior = torch.randn(N, K, k, k, generator=g).to(gpu_device)
    for acc in (0, 1):
        out = prior.clone()
        check(lib.hg_demod_weight_term(w.data_ptr(), gd.data_ptr(), d.data_ptr(), s1.data_ptr(), out.data_ptr(), B, N, K,
                                       k * k, acc, raw_stream(gpu_device)), 'hg_demod_weight_term')
        want = ref + prior.double() if acc else ref
        assert relmax(out.cpu().numpy(), want.cpu().numpy()) <= 2e-6, acc
    assert lib.hg_demod_weight_term(None, gd.data_ptr(), d.data_ptr(), s1.data_ptr(), prior.data_ptr(), B, N, K, k * k, 0,
                                    raw_stream(gpu_device)) != 0


@pytest.mark.parametrize('B,N,K', [(32, 64, 48), (4, 5, 3), (32, 2048, 2048), (2, 1024, 2048), (40, 130, 70), (3, 4100, 65)])
def test_demod_style_grad_kernel_matches_fp64(B, N, K, gpu_device):
    """hg_demod_style_grad (the demodulation coefficient's style gradient) against the aten formula in fp64; B above the
    kernel's 32-row pass, N above one 64-row split table, ragged K."""
    from histogan_amd._lib import check, lib, raw_stream
    g = torch.Generator(device='cpu').manual_seed(3 * B + 7 * N + 13 * K)
    wsq = (torch.rand(N, K, generator=g) / K).to(gpu_device)
    s1 = (torch.randn(B, K, generator=g) * 0.5 + 1.0).to(gpu_device)
    gd = torch.randn(B, N, generator=g).to(gpu_device)
    d = torch.rsqrt((s1 * s1) @ wsq.t() + 1e-8)
    ref = 2.0 * s1.double() * ((gd.double() * (-0.5) * d.double() ** 3) @ wsq.double())
    gy = torch.full((B, K), float('nan'), device=gpu_device)
    nb = lib.hg_demod_style_grad_workspace_bytes(B, N, K)
    ws = torch.empty(nb, dtype=torch.uint8, device=gpu_device)
    args = (gd.data_ptr(), d.data_ptr(), s1.data_ptr(), wsq.data_ptr(), gy.data_ptr(), B, N, K, ws.data_ptr())
    check(lib.hg_demod_style_grad(*args, nb, raw_stream(gpu_device)), 'hg_demod_style_grad')
    assert relmax(gy.cpu().numpy(), ref.cpu().numpy()) <= 2e-6
    first = gy.clone()
    check(lib.hg_demod_style_grad(*args, nb, raw_stream(gpu_device)), 'hg_demod_style_grad')
    assert torch.equal(first, gy)                                   # fixed summation order
    assert lib.hg_demod_style_grad(*args, nb - 4, raw_stream(gpu_device)) != 0      # workspace too small
