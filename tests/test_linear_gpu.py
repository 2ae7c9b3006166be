"""Grouped linear layers (include/hg_linear.h: the generator's style projections, histoGAN/histoGAN.py:372, 450, 454, as one
launch per pass) against F.linear in fp64: outputs, input / weight / bias gradients, determinism, and the generator using
them == the generator with one library GEMM per projection."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import relmax

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,K,Ns,groups', [
    (32, 512, [64, 2048, 2048, 2048, 1024, 1024, 32, 32, 32], [0, 0, 0, 1, 1, 1, 2, 2, 2]),   # C3-like: three per block
    (1, 512, [64, 36], [0, 0]), (8, 64, [128, 4, 260], [0, 1, 1]), (33, 96, [100, 32], [0, 1]), (64, 512, [2048, 512, 12], [0, 0, 1]),
])
def test_grouped_linear_matches_fp64(B, K, Ns, groups, gpu_device):
    from histogan_amd import ops
    torch.manual_seed(B + K + len(Ns))
    dev = gpu_device
    G = max(groups) + 1
    xs = [torch.randn(B, K, device=dev, requires_grad=True) for _ in range(G)]
    layers = [nn.Linear(K, n).to(dev) for n in Ns]
    assert ops.grouped_linear_supported(xs, [m.weight for m in layers])
    ys = ops.grouped_linear(xs, layers, groups)
    gos = [torch.randn_like(y) for y in ys]
    params = [p for m in layers for p in (m.weight, m.bias)]
    grads = torch.autograd.grad(ys, xs + params, gos)
    xd = [x.detach().double().requires_grad_(True) for x in xs]
    pd = [p.detach().double().requires_grad_(True) for p in params]
    yr = [F.linear(xd[g], pd[2 * i], pd[2 * i + 1]) for i, g in enumerate(groups)]
    gr = torch.autograd.grad(yr, xd + pd, [g.double() for g in gos])
    for a, b in zip(ys, yr):
        assert a.shape == b.shape and relmax(a.detach().cpu().numpy(), b.detach().cpu().numpy()) <= 2e-6
    for a, b in zip(grads, gr):
        assert a.shape == b.shape and relmax(a.cpu().numpy(), b.cpu().numpy()) <= 2e-6
    # deterministic: bit-identical repeats (fixed-order slab sums, no atomics)
    ys2 = ops.grouped_linear(xs, layers, groups)
    grads2 = torch.autograd.grad(ys2, xs + params, gos)
    assert all(torch.equal(a, b) for a, b in zip(ys, ys2)) and all(torch.equal(a, b) for a, b in zip(grads, grads2))


def test_generator_with_grouped_projections_equals_per_layer_gemms(gpu_device):
    """Generator.forward with the grouped launch vs. HG_GROUPED_STYLES=0's path (one F.linear per projection): same rgb and
    gradients up to fp32 summation order."""
    from histoGAN import Generator
    from histogan_amd import ops
    torch.manual_seed(5)
    dev, B, S_, LAT, CAP = gpu_device, 3, 64, 512, 4
    Gn = Generator(S_, LAT, network_capacity=CAP).to(dev)
    L = Gn.num_layers
    styles = torch.randn(B, L - 2, LAT, device=dev, requires_grad=True)
    hists = torch.randn(B, 2, LAT, device=dev, requires_grad=True)
    noise = torch.rand(B, S_, S_, 1, device=dev)
    go = torch.randn(B, 3, S_, S_, device=dev)
    names = [n for n, _ in Gn.named_parameters()]
    params = [p for _, p in Gn.named_parameters()]

    def run():
        rgb = Gn(styles, hists, noise)
        return rgb.detach(), torch.autograd.grad(rgb, [styles, hists] + params, go)

    assert ops.GROUPED_STYLES
    a_rgb, a_gr = run()
    ops.GROUPED_STYLES = False
    try:
        b_rgb, b_gr = run()
    finally:
        ops.GROUPED_STYLES = True
    assert relmax(a_rgb.cpu().numpy(), b_rgb.cpu().numpy()) <= 1e-5
    for n, a, b in zip(['styles', 'hists'] + names, a_gr, b_gr):
        assert relmax(a.cpu().numpy(), b.cpu().numpy()) <= 1e-4, n
