"""Dense contraction of the HistoGAN networks on the hand-written fp32-MFMA kernels of include/hg_conv.h.

`conv2d_same(x, w, bias=None)` == `F.conv2d(x, w, bias, padding=k//2)` for k in {1, 3}, stride 1 -- the
contraction inside Conv2DMod.forward (histoGAN/histoGAN.py:431-439, executed with the shared weight on
modulated activations) and the discriminator's stride-1 convolutions (:510-515).  Output, data gradient and
weight gradient are three launches of the implicit-GEMM kernels (k_conv with forward-/dgrad-packed
weights, k_wgrad); first-order differentiable.
"""
import ctypes

import torch

from ._lib import check, lib

PACK_FWD, PACK_DGRAD = 0, 1


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32c(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def pack_weights(w, mode):
    """(Co,Ci,k,k) -> the packed operand Wt[k*k][Kp][Np] of hg_conv2d_same (mode PACK_FWD / PACK_DGRAD)."""
    Co, Ci, k, _ = w.shape
    n = lib.hg_conv_packed_elems(Co, Ci, k, mode)
    if n == 0:
        raise ValueError(f'conv weights {tuple(w.shape)}: only square 1x1 / 3x3 kernels are implemented')
    with torch.cuda.device(w.device):
        wt = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib.hg_conv_pack_weights(w.data_ptr(), wt.data_ptr(), Co, Ci, k, mode, _st(w)), 'hg_conv_pack_weights')
    return wt


def conv_packed(x, wt, N, ksize, iscale=None, oscale=None, bias=None):
    """out[b,n] = oscale[b,n] * sum_k conv(iscale[b,k] * x[b,k], Wt[.,k,n]) + bias[n]   (x: (B,K,H,W) contiguous)."""
    B, K, H, W = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty((B, N, H, W), dtype=torch.float32, device=x.device)
        check(lib.hg_conv2d_same(x.data_ptr(), wt.data_ptr(), out.data_ptr(), _ptr(iscale), _ptr(oscale), _ptr(bias),
                                 B, K, N, H, W, ksize, _st(x)), 'hg_conv2d_same')
    return out


def conv_wgrad(x, gout, ksize, iscale=None, gscale=None):
    """gw[n,k,dy,dx] = sum_{b,y,x} gscale[b,n] gout[b,n,y,x] * iscale[b,k] x[b,k,y+dy-p,x+dx-p]."""
    B, K, H, W = x.shape
    N = gout.shape[1]
    with torch.cuda.device(x.device):
        nbytes = lib.hg_conv2d_wgrad_workspace_bytes(B, K, N, H, W, ksize)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=x.device)
        gw = torch.empty((N, K, ksize, ksize), dtype=torch.float32, device=x.device)
        check(lib.hg_conv2d_wgrad(x.data_ptr(), gout.data_ptr(), gw.data_ptr(), _ptr(iscale), _ptr(gscale),
                                  B, K, N, H, W, ksize, ws.data_ptr(), ws.numel(), _st(x)), 'hg_conv2d_wgrad')
    return gw


class _Conv2dSame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        if not x.is_cuda:
            raise RuntimeError(f'conv2d_same: tensor on {x.device}; the MI355X-native path has no CPU implementation')
        x, w = _f32c(x.detach()), _f32c(w.detach())
        b = None if bias is None else _f32c(bias.detach())
        Co, Ci, k, k2 = w.shape
        if k != k2 or k not in (1, 3) or x.shape[1] != Ci:
            raise ValueError(f'conv2d_same: x {tuple(x.shape)} / w {tuple(w.shape)} not supported (1x1 / 3x3, stride 1)')
        out = conv_packed(x, pack_weights(w, PACK_FWD), Co, k, bias=b)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _f32c(g.detach())
        Co, Ci, k, _ = w.shape
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = conv_packed(g, pack_weights(w, PACK_DGRAD), Ci, k)
        if ctx.needs_input_grad[1]:
            gw = conv_wgrad(x, g, k)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(dim=(0, 2, 3))
        return gx, gw, gb


def conv2d_same(x, w, bias=None):
    """F.conv2d(x, w, bias, stride=1, padding=k//2) for k in {1,3} on the MFMA implicit-GEMM kernels."""
    return _Conv2dSame.apply(x, w, bias)
