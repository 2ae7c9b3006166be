"""autograd Functions over the generator kernels of include/hg_nets.h (first-order differentiable).

modulate            [bilinear x2 ->] x*(s+1)             prologue of Conv2DMod   (histoGAN/histoGAN.py:420-424, 447-448)
demod_noise_lrelu   lrelu(conv*d + noise)                epilogue               (histoGAN/histoGAN.py:427-429, 465-476)
upsample2x          nn.Upsample(bilinear, x2) of the RGB skip                   (histoGAN/histoGAN.py:377-378)
"""
import ctypes

import torch

from ._lib import check, lib


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32c(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def _ws(t, B, C, H, W):
    n = lib.hg_nets_workspace_bytes(B, C, H, W)
    return torch.empty(max(n, 4), dtype=torch.uint8, device=t.device), n


def channel_sum(g):
    """(B, C, H, W) -> (C): sum over batch and pixels (bias gradient), deterministic two-stage reduction."""
    g = _f32c(g.detach())
    B, C, H, W = g.shape
    with torch.cuda.device(g.device):
        out = torch.empty(C, dtype=torch.float32, device=g.device)
        ws, n = _ws(g, B, C, H, W)
        check(lib.hg_channel_sum(g.data_ptr(), out.data_ptr(), B, C, H * W, ws.data_ptr(), n, _st(g)), 'hg_channel_sum')
    return out


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: tensor on {t.device}; the MI355X-native path has no CPU implementation')


class _Modulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, upsample):
        _need_gpu(x, 'modulate')
        x = _f32c(x.detach())
        s = None if s is None else _f32c(s.detach())
        B, C, H, W = x.shape
        f = 2 if upsample else 1
        with torch.cuda.device(x.device):
            out = torch.empty((B, C, H * f, W * f), dtype=torch.float32, device=x.device)
            check(lib.hg_modulate_fwd(x.data_ptr(), None if s is None else s.data_ptr(), out.data_ptr(),
                                      B, C, H, W, int(upsample), _st(x)), 'hg_modulate_fwd')
        ctx.save_for_backward(x, s)
        ctx.upsample = bool(upsample)
        return out

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = _f32c(g.detach())
        B, C, H, W = x.shape
        with torch.cuda.device(x.device):
            gx = torch.empty_like(x)
            gs = None if s is None else torch.empty_like(s)
            ws, n = _ws(x, B, C, H, W)
            check(lib.hg_modulate_bwd(g.data_ptr(), x.data_ptr(), None if s is None else s.data_ptr(),
                                      gx.data_ptr(), None if gs is None else gs.data_ptr(), B, C, H, W,
                                      int(ctx.upsample), ws.data_ptr(), n, _st(x)), 'hg_modulate_bwd')
        return gx, gs, None


def modulate(x, s, upsample=False):
    """(B,C,H,W), (B,C) -> [up2](x) * (s+1)[:, :, None, None]."""
    return _Modulate.apply(x, s, upsample)


def upsample2x(x):
    """Bilinear x2 (align_corners=False, edge clamp) == nn.Upsample(scale_factor=2, mode='bilinear')."""
    return _Modulate.apply(x, None, True)


class _DemodNoiseLrelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, conv, d, nzt, wn, bn):
        _need_gpu(conv, 'demod_noise_lrelu')
        conv = _f32c(conv.detach())
        d = None if d is None else _f32c(d.detach())
        nzt, wn, bn = _f32c(nzt.detach()), _f32c(wn.detach().reshape(-1)), _f32c(bn.detach())
        B, O, H, W = conv.shape
        if H != W:
            raise ValueError('square feature maps only (the noise permute of the reference needs H == W)')
        S = nzt.shape[-1]
        with torch.cuda.device(conv.device):
            out = torch.empty_like(conv)
            check(lib.hg_demod_noise_lrelu_fwd(conv.data_ptr(), None if d is None else d.data_ptr(), nzt.data_ptr(),
                                               wn.data_ptr(), bn.data_ptr(), out.data_ptr(), B, O, H, S, _st(conv)),
                  'hg_demod_noise_lrelu_fwd')
        ctx.save_for_backward(conv, d, nzt, out)
        ctx.wn_shape = None
        return out

    @staticmethod
    def backward(ctx, g):
        conv, d, nzt, out = ctx.saved_tensors
        g = _f32c(g.detach())
        B, O, H, _ = conv.shape
        S = nzt.shape[-1]
        with torch.cuda.device(conv.device):
            gconv = torch.empty_like(conv)
            gd = None if d is None else torch.empty_like(d)
            gw = torch.empty((B, O), dtype=torch.float32, device=conv.device)
            gb = torch.empty((B, O), dtype=torch.float32, device=conv.device)
            ws, n = _ws(conv, B, O, H, H)
            check(lib.hg_demod_noise_lrelu_bwd(g.data_ptr(), out.data_ptr(), conv.data_ptr(),
                                               None if d is None else d.data_ptr(), nzt.data_ptr(), gconv.data_ptr(),
                                               None if gd is None else gd.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                               B, O, H, S, ws.data_ptr(), n, _st(conv)), 'hg_demod_noise_lrelu_bwd')
        return gconv, gd, None, gw.sum(0).reshape(-1, 1), gb.sum(0)


def demod_noise_lrelu(conv, d, nzt, wn, bn):
    """lrelu_0.2(conv * d[:, :, None, None] + wn[o] * nzt[b, i, j] + bn[o]);  wn: Linear(1,O).weight (O,1)."""
    return _DemodNoiseLrelu.apply(conv, d, nzt, wn, bn)
