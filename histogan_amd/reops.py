"""autograd Functions over the ReHistoGAN kernels of include/hg_recolor.h.

instnorm_lrelu   nn.InstanceNorm2d(affine=False) + LeakyReLU(0.2)   EncoderBlock.net  (ReHistoGAN/rehistoGAN.py:489-496)
stencil3         F.conv2d(x, taps.expand(1,C,3,3), padding=1)        sobel_op / laplacian_op (:235-256)
gaussian_valid   depthwise 15x15 filter without padding             gaussian_op (:228-232)

stencil3 and gaussian_valid are linear, their backward is the adjoint kernel wrapped as a Function again, so both
are differentiable to any order; instnorm_lrelu is first-order (nothing on the path differentiates it twice: the
gradient penalty runs on D(real images) only, :136-144).
"""
import ctypes

import torch

from ._lib import check, lib, on_device, raw_stream
from .ops import _f32c, _need_gpu, _st


def _in_ws(t, P):
    n = lib.hg_instnorm_workspace_bytes(P)
    return torch.empty(max(n, 4), dtype=torch.uint8, device=t.device), n


class _InstNormLrelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, slope):
        _need_gpu(x, 'instnorm_lrelu')
        x = _f32c(x.detach())
        B, C, H, W = x.shape
        with on_device(x.device):
            out = torch.empty_like(x)
            stats = torch.empty((B * C, 2), dtype=torch.float32, device=x.device)
            ws, n = _in_ws(x, B * C)
            check(lib.hg_instnorm_lrelu_fwd(x.data_ptr(), out.data_ptr(), stats.data_ptr(), B * C, H * W, eps, slope,
                                            ws.data_ptr(), n, _st(x)), 'hg_instnorm_lrelu_fwd')
        ctx.save_for_backward(out, stats)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, g):
        out, stats = ctx.saved_tensors
        g = _f32c(g.detach())
        B, C, H, W = out.shape
        with on_device(out.device):
            gx = torch.empty_like(out)
            ws, n = _in_ws(out, B * C)
            check(lib.hg_instnorm_lrelu_bwd(g.data_ptr(), out.data_ptr(), stats.data_ptr(), gx.data_ptr(), B * C, H * W,
                                            ctx.slope, ws.data_ptr(), n, _st(out)), 'hg_instnorm_lrelu_bwd')
        return gx, None, None


def instnorm_lrelu(x, eps=1e-5, slope=0.2):
    """(B,C,H,W) -> lrelu((x - mean_bc) / sqrt(var_bc + eps)), biased variance over H*W."""
    return _InstNormLrelu.apply(x, float(eps), float(slope))


def _stencil_raw(x, taps, C, adjoint):
    _need_gpu(x, 'stencil3')
    x = _f32c(x.detach())
    B, _, H, W = x.shape
    arr = (ctypes.c_float * 9)(*[float(v) for v in taps])
    with on_device(x.device):
        out = torch.empty((B, C if adjoint else 1, H, W), dtype=torch.float32, device=x.device)
        check(lib.hg_stencil3(x.data_ptr(), out.data_ptr(), arr, B, C, H, W, int(adjoint), _st(x)), 'hg_stencil3')
    return out


class _Stencil3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, taps, C, adjoint):
        ctx.taps, ctx.C, ctx.adjoint = taps, C, adjoint
        return _stencil_raw(x, taps, C, adjoint)

    @staticmethod
    def backward(ctx, g):
        return _Stencil3.apply(g, ctx.taps, ctx.C, not ctx.adjoint), None, None, None


def stencil3(x, taps):
    """x (B,C,H,W), taps: 9 floats (row-major 3x3) -> (B,1,H,W): the stencil summed over the C channels, zero padding 1."""
    taps = tuple(float(v) for v in torch.as_tensor(taps, dtype=torch.float32).reshape(-1).tolist())
    if len(taps) != 9:
        raise ValueError('stencil3: 9 taps expected')
    return _Stencil3.apply(x, taps, x.shape[1], False)


def _dw_raw(x, k, H, W, adjoint):
    _need_gpu(x, 'gaussian_valid')
    x = _f32c(x.detach())
    B, C = x.shape[:2]
    KS = k.shape[-1]
    with on_device(x.device):
        shape = (B, C, H, W) if adjoint else (B, C, H - KS + 1, W - KS + 1)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        check(lib.hg_depthwise_valid(x.data_ptr(), k.data_ptr(), out.data_ptr(), B * C, H, W, KS, int(adjoint), _st(x)),
              'hg_depthwise_valid')
    return out


class _Depthwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, H, W, adjoint):
        ctx.k, ctx.H, ctx.W, ctx.adjoint = k, H, W, adjoint
        return _dw_raw(x, k, H, W, adjoint)

    @staticmethod
    def backward(ctx, g):
        return _Depthwise.apply(g, ctx.k, ctx.H, ctx.W, not ctx.adjoint), None, None, None, None


def gaussian_valid(x, kernel):
    """x (B,C,H,W); kernel: the (KS,KS) filter applied to EVERY channel (the reference repeats one Gaussian over the
    channels, :216-217), or the reference's (C,1,KS,KS) depthwise weight whose channel slices are all equal.
    -> (B,C,H-KS+1,W-KS+1), no padding."""
    k = kernel.detach()
    if k.dim() == 4:
        k = k[0, 0]
    k = _f32c(k).to(x.device)
    return _Depthwise.apply(x, k, x.shape[2], x.shape[3], False)
