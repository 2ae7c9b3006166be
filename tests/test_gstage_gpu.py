"""hg_gstage_bwd (include/hg_nets.h): the fused backward between two generator convolutions against fp64 autograd of the
chain it replaces (GeneratorBlock.forward / Conv2DMod / RGBBlock, histoGAN/histoGAN.py:461-479, 420-440, 380-390), and the
one-node generator (histogan_amd/gfused.py) against the per-block autograd path."""
import pytest
import torch
import torch.nn.functional as F

from conftest import relmax

pytestmark = pytest.mark.gpu

CASES = [
    # B, C, H, S, up, rgb
    (2, 8, 4, 16, True, True),        # 4 lanes per plane, many planes per block
    (3, 20, 8, 32, True, True),
    (2, 6, 16, 16, True, True),       # 64 lanes per plane (UP: 128 px pairs -> 128 lanes)
    (2, 5, 32, 64, True, True),       # one plane per block
    (1, 3, 128, 128, True, True),     # chunks > 1
    (2, 8, 4, 16, False, False),      # conv2 -> conv1 of a block
    (3, 12, 8, 8, False, False),
    (2, 4, 64, 64, False, False),
    (2, 7, 16, 32, False, True),      # same resolution + to-RGB (not used by the generator; the kernel serves it)
    (2, 16, 32, 32, None, True),      # last block: to-RGB only
    (1, 2, 256, 256, None, True),
    (2, 5, 32, 64, True, False),      # next convolution only
]


@pytest.mark.parametrize('B,Cc,H,S,up,rgb', CASES)
def test_gstage_bwd_matches_fp64_autograd(B, Cc, H, S, up, rgb, gpu_device):
    from histogan_amd.gfused import gstage_bwd
    g = torch.Generator().manual_seed(B * 1000 + Cc * 10 + H)
    dev = gpu_device
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    conv, d = rnd(B, Cc, H, H), torch.rand(B, Cc, generator=g, dtype=torch.float64) + 0.5
    nzt, wn, bn = torch.rand(B, S, S, generator=g, dtype=torch.float64), rnd(Cc) * 0.5, rnd(Cc) * 0.2
    sa, srgb, w = rnd(B, Cc) * 0.5, rnd(B, Cc) * 0.5, rnd(3, Cc)
    has_a = up is not None
    ga = rnd(B, Cc, 2 * H, 2 * H) if up else (rnd(B, Cc, H, H) if has_a else None)
    g_rgb = rnd(B, 3, H, H) if rgb else None
    leaves = [t.requires_grad_(True) for t in (conv, d, wn, bn, sa, srgb, w)]
    pre = conv * d[:, :, None, None] + wn[None, :, None, None] * nzt[:, None, :H, :H] + bn[None, :, None, None]
    out = F.leaky_relu(pre, 0.2)
    loss = 0.0
    if has_a:
        xa = F.interpolate(out, scale_factor=2, mode='bilinear', align_corners=False) if up else out
        loss = loss + (xa * (sa + 1)[:, :, None, None] * ga).sum()
    if rgb:
        loss = loss + (torch.einsum('kc,bcij->bkij', w, out * (srgb + 1)[:, :, None, None]) * g_rgb).sum()
    want = torch.autograd.grad(loss, leaves, allow_unused=True)
    f = lambda t: None if t is None else t.detach().float().to(dev).contiguous()
    gconv, gs_a, gs_rgb, gw_rgb, gd, gwn, gbn = gstage_bwd(f(out), f(ga), f(sa) if has_a else None, bool(up), f(g_rgb),
                                                           f(w) if rgb else None, f(srgb) if rgb else None, f(d), f(nzt),
                                                           f(wn), f(bn))
    torch.cuda.synchronize()
    tol = 2e-5
    assert relmax(gconv.cpu().numpy(), want[0].numpy()) <= tol
    assert relmax(gd.cpu().numpy(), want[1].numpy()) <= tol
    assert relmax(gwn.cpu().numpy(), want[2].numpy()) <= tol
    assert relmax(gbn.cpu().numpy(), want[3].numpy()) <= tol
    if has_a:
        assert relmax(gs_a.cpu().numpy(), want[4].numpy()) <= tol
    if rgb:
        assert relmax(gs_rgb.cpu().numpy(), want[5].numpy()) <= tol
        assert relmax(gw_rgb.cpu().numpy(), want[6].numpy()) <= tol


def test_gstage_bwd_refuses_bad_arguments(gpu_device):
    from histogan_amd._lib import lib
    t = torch.zeros(2, 4, 8, 8, device=gpu_device)
    v = torch.zeros(64, device=gpu_device)
    p = lambda x: x.data_ptr()
    ws = torch.zeros(1 << 16, dtype=torch.uint8, device=gpu_device)
    # no upstream gradient at all
    assert lib.hg_gstage_bwd(p(t), None, None, 0, None, None, None, 0, p(v), p(t), p(v), p(v), 8, p(t), None, None, None, p(v),
                             p(v), p(v), 2, 4, 8, p(ws), ws.numel(), None) == -1
    # workspace too small
    assert lib.hg_gstage_bwd(p(t), p(t), p(v), 0, None, None, None, 0, p(v), p(t), p(v), p(v), 8, p(t), p(v), None, None, p(v),
                             p(v), p(v), 2, 4, 8, p(ws), 4, None) != 0


@pytest.mark.parametrize('size,cap,B', [(32, 2, 3), (64, 4, 2)])
def test_fused_generator_node_equals_per_block_autograd(size, cap, B, gpu_device):
    """The same network, styles, noise and upstream gradient through (a) the one-node training pass and (b) the per-block
    autograd path: output and every gradient agree to fp32 rounding (different summation orders only)."""
    from histogan_amd import gfused
    from histogan_amd.nets import Generator
    torch.manual_seed(3)
    G = Generator(size, 64, cap).to(gpu_device)
    for b in G.blocks:                      # (the reference initialises the noise projections to zero: make them count)
        for m in (b.to_noise1, b.to_noise2):
            torch.nn.init.normal_(m.weight, std=0.3)
            torch.nn.init.normal_(m.bias, std=0.1)
    L = G.num_layers
    styles = torch.randn(B, L - 2, 64, device=gpu_device, requires_grad=True)
    hists = torch.randn(B, 2, 64, device=gpu_device, requires_grad=True)
    noise = torch.rand(B, size, size, 1, device=gpu_device)
    gout = torch.randn(B, 3, size, size, device=gpu_device)
    res = {}
    for mode in (True, False):
        gfused.GFUSED = mode
        try:
            rgb = G(styles, hists, noise)
            grads = torch.autograd.grad((rgb * gout).sum(), [styles, hists] + list(G.parameters()))
        finally:
            gfused.GFUSED = True
        res[mode] = (rgb.detach(), grads)
    assert relmax(res[True][0].cpu().numpy(), res[False][0].cpu().numpy()) <= 1e-6
    # the no-autograd twin (gfused.generator_infer: in-kernel modulation, demodulation coefficients ahead on a second stream)
    with torch.no_grad():
        inf = G(styles, hists, noise)
        gfused.GFUSED = False
        try:
            inf_ref = G(styles, hists, noise)
        finally:
            gfused.GFUSED = True
    torch.cuda.synchronize()
    assert relmax(inf.cpu().numpy(), inf_ref.cpu().numpy()) <= 1e-6
    assert relmax(inf.cpu().numpy(), res[False][0].cpu().numpy()) <= 2e-6
    names = ['styles', 'hists'] + [n for n, _ in G.named_parameters()]
    for n, a, b_ in zip(names, res[True][1], res[False][1]):
        assert relmax(a.cpu().numpy(), b_.cpu().numpy()) <= 2e-5, n
