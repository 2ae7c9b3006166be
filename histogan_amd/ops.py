"""autograd Functions over the generator kernels of include/hg_nets.h (first-order differentiable).

modulate            [bilinear x2 ->] x*(s+1)             prologue of Conv2DMod   (histoGAN/histoGAN.py:420-424, 447-448)
demod_noise_lrelu   lrelu(conv*d + noise)                epilogue               (histoGAN/histoGAN.py:427-429, 465-476)
upsample2x          nn.Upsample(bilinear, x2) of the RGB skip                   (histoGAN/histoGAN.py:377-378)
"""
import os
import ctypes

import torch

from ._lib import check, lib, on_device, raw_stream


KEEP_CONV = os.environ.get('HG_DNL_KEEP_CONV', '1') != '0'
SKINNY_SPLIT = os.environ.get('HG_SKINNY_SPLIT', '1') != '0'   # _skinny_mm: chunked bmm + sum (0: plain mm)
FUSED_DEMOD_BWD = os.environ.get('HG_FUSED_DEMOD_BWD', '1') != '0'   # hg_demod_style_grad (0: aten ops)


def _st(t):
    return raw_stream(t.device)


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
    with on_device(g.device):
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
        with on_device(x.device):
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
        with on_device(x.device):
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
        with on_device(conv.device):
            out = torch.empty_like(conv)
            check(lib.hg_demod_noise_lrelu_fwd(conv.data_ptr(), None if d is None else d.data_ptr(), nzt.data_ptr(),
                                               wn.data_ptr(), bn.data_ptr(), out.data_ptr(), B, O, H, S, _st(conv)),
                  'hg_demod_noise_lrelu_fwd')
        # HG_DNL_KEEP_CONV=0: the convolution output is not kept for the backward; conv*d is recovered from `out` there
        # (pre = out > 0 ? out : 5 out, minus the noise term) -- one tensor less per stage and one read less in k_dnl_bwd.
        # Measured at C3: 54.7 vs 54.5 ms per plain step (the kernel hides behind the side-stream weight gradients), so
        # the stored operand (no recovery rounding) stays the default.
        if KEEP_CONV:
            ctx.save_for_backward(conv, d, nzt, out, wn, bn)
        else:
            ctx.save_for_backward(None, d, nzt, out, wn, bn)
        return out

    @staticmethod
    def backward(ctx, g):
        conv, d, nzt, out, wn, bn = ctx.saved_tensors
        g = _f32c(g.detach())
        B, O, H, _ = out.shape
        S = nzt.shape[-1]
        with on_device(out.device):
            gconv = torch.empty_like(out)
            gd = None if d is None else torch.empty_like(d)
            gw = torch.empty((B, O), dtype=torch.float32, device=out.device)
            gb = torch.empty((B, O), dtype=torch.float32, device=out.device)
            ws, n = _ws(out, B, O, H, H)
            check(lib.hg_demod_noise_lrelu_bwd(g.data_ptr(), out.data_ptr(), None if conv is None else conv.data_ptr(),
                                               None if d is None else d.data_ptr(), nzt.data_ptr(),
                                               None if conv is not None else wn.data_ptr(),
                                               None if conv is not None else bn.data_ptr(), gconv.data_ptr(),
                                               None if gd is None else gd.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                               B, O, H, S, ws.data_ptr(), n, _st(out)), 'hg_demod_noise_lrelu_bwd')
        return gconv, gd, None, gw.sum(0).reshape(-1, 1), gb.sum(0)


def demod_noise_lrelu(conv, d, nzt, wn, bn):
    """lrelu_0.2(conv * d[:, :, None, None] + wn[o] * nzt[b, i, j] + bn[o]);  wn: Linear(1,O).weight (O,1)."""
    return _DemodNoiseLrelu.apply(conv, d, nzt, wn, bn)


TORGB = os.environ.get('HG_TORGB', '1') != '0'      # the to-RGB path as one stream over x per direction (hg_torgb_fwd / _bwd)


def torgb_supported(x, weight):
    return (TORGB and x.is_cuda and x.dtype == torch.float32 and weight.shape[2] == 1 and weight.shape[3] == 1
            and weight.shape[0] <= 4 and (x.shape[2] * x.shape[3]) % 4 == 0 and weight.shape[0] * weight.shape[1] * 4 <= 48 * 1024)


class _ToRGB(torch.autograd.Function):
    """rgb = conv1x1(x * (style + 1), W) [+ prev]: RGBBlock's modulated convolution without demodulation plus the running RGB
    image (histoGAN/histoGAN.py:380-390) as ONE pass over x forward (hg_torgb_fwd) and ONE backward (hg_torgb_bwd: gx, the
    style gradient and the weight gradient from the same read of x) -- instead of a modulated copy of x, a 3-row matrix
    launch and a residual add forward, and data gradient + weight gradient + modulation adjoint backward.  First order."""

    @staticmethod
    def forward(ctx, x, style, weight, prev):
        x_, s_, w_ = _f32c(x.detach()), _f32c(style.detach()), _f32c(weight.detach())
        B, O, H, W = x_.shape
        C = w_.shape[0]
        p_ = None if prev is None else _f32c(prev.detach())
        with on_device(x_.device):
            out = torch.empty((B, C, H, W), dtype=torch.float32, device=x_.device)
            check(lib.hg_torgb_fwd(x_.data_ptr(), s_.data_ptr(), w_.data_ptr(), None if p_ is None else p_.data_ptr(),
                                   out.data_ptr(), B, O, C, H * W, _st(x_)), 'hg_torgb_fwd')
        ctx.save_for_backward(x_, s_, w_)
        ctx.has_prev = prev is not None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x_, s_, w_ = ctx.saved_tensors
        g = _f32c(g.detach())
        B, O, H, W = x_.shape
        C = w_.shape[0]
        with on_device(x_.device):
            gx = torch.empty_like(x_)
            gs = torch.empty_like(s_)
            gw = torch.empty_like(w_)
            nb = lib.hg_torgb_bwd_workspace_bytes(B, O, C, H * W)
            ws = torch.empty(nb, dtype=torch.uint8, device=x_.device)
            check(lib.hg_torgb_bwd(g.data_ptr(), x_.data_ptr(), s_.data_ptr(), w_.data_ptr(), gx.data_ptr(), gs.data_ptr(),
                                   gw.data_ptr(), B, O, C, H * W, ws.data_ptr(), nb, _st(x_)), 'hg_torgb_bwd')
        return gx, gs, gw, (g if ctx.has_prev else None)


def torgb(x, style, weight, prev=None):
    """conv1x1(x * (style + 1), weight) + prev in one launch per direction -- see _ToRGB."""
    return _ToRGB.apply(x, style, weight, prev)


FUSED_DNL = os.environ.get('HG_FUSED_DNL', '1') != '0'   # conv + demodulation + noise + LeakyReLU as one forward launch in training


class _ConvDnl(torch.autograd.Function):
    """out = lrelu_0.2(d[b,o] * conv(xm, W) + wn[o] * nzt[b,i,j] + bn[o]) on the already MODULATED input xm, as ONE launch
    (the fused epilogue of hg_wino_conv2d / hg_modconv2d_fwd): the training forward of a generator stage
    (histoGAN/histoGAN.py:431-439, 465-476) without the separate k_dnl_fwd pass and without storing the convolution output.
    Backward: k_dnl_bwd recovers conv * d from `out` (conv = NULL form), then the data gradient and the weight gradient
    (xm is materialised, so the weight gradient needs no modulation of its own -- the cost that made the fully fused stage
    lose, _ModConvStage).  First order only, like _DemodNoiseLrelu."""

    @staticmethod
    def forward(ctx, xm, w, d, nzt, wn, bn):
        from . import conv as C
        _need_gpu(xm, 'conv_dnl')
        xm_, w_ = _f32c(xm.detach()), _f32c(w.detach())
        d_ = None if d is None else _f32c(d.detach())
        nzt_, wn_, bn_ = _f32c(nzt.detach()), _f32c(wn.detach().reshape(-1)), _f32c(bn.detach())
        N, k = w_.shape[0], w_.shape[2]
        if xm_.shape[2] != xm_.shape[3]:
            raise ValueError('square feature maps only (the noise permute of the reference needs H == W)')
        out = C.modconv_fwd_packed(xm_, C.pack_weights(w_, C.PACK_FWD), N, k, None, d_, bn_, wn_, nzt_, nzt_.shape[-1], 0.2)
        ctx.save_for_backward(xm, w, d_, nzt_, out, wn_, bn_)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import conv as C
        xm, w, d, nzt_, out, wn_, bn_ = ctx.saved_tensors
        g = _f32c(g.detach())
        B, N, H, _ = out.shape
        K, k = w.shape[1], w.shape[2]
        S = nzt_.shape[-1]
        with on_device(out.device):
            gconv = torch.empty_like(out)
            gd = None if d is None else torch.empty_like(d)
            gw_p = torch.empty((B, N), dtype=torch.float32, device=out.device)
            gb_p = torch.empty((B, N), dtype=torch.float32, device=out.device)
            ws, n = _ws(out, B, N, H, H)
            check(lib.hg_demod_noise_lrelu_bwd(g.data_ptr(), out.data_ptr(), None, None if d is None else d.data_ptr(),
                                               nzt_.data_ptr(), wn_.data_ptr(), bn_.data_ptr(), gconv.data_ptr(),
                                               None if gd is None else gd.data_ptr(), gw_p.data_ptr(), gb_p.data_ptr(),
                                               B, N, H, S, ws.data_ptr(), n, _st(out)), 'hg_demod_noise_lrelu_bwd')
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = C.conv_dgrad_packed(gconv, C.pack_weights(_f32c(w.detach()), C.PACK_DGRAD), K, xm.shape[2], xm.shape[3], k)
        if ctx.needs_input_grad[1] and not C._skip_wgrad and not C._direct_wgrad(w, xm, gconv, 1):
            gw = C.conv_wgrad(_f32c(xm.detach()), gconv, k)
        return gx, gw, gd, None, gw_p.sum(0).reshape(-1, 1), gb_p.sum(0)


def conv_dnl(xm, w, d, nzt, wn, bn):
    """lrelu_0.2(d * conv(xm, w) + wn * nzt + bn) in one forward launch -- see _ConvDnl."""
    return _ConvDnl.apply(xm, w, d, nzt, wn, bn)


class _ModConvStage(torch.autograd.Function):
    """One generator convolution stage as a single forward launch (hg_modconv2d_fwd):

        out = act( d[b,o] * conv(up?(x) * (style+1), W) + wn[o] * nzt[b,i,j] + bn[o] ),   d = demodulation

    == Conv2DMod.forward + noise add + LeakyReLU(0.2) of GeneratorBlock.forward (histoGAN/histoGAN.py:420-440,
    465-476); with act=False / no noise it is the to-RGB convolution (:375, 383).  The modulated input, the
    convolution output before demodulation and the pre-activation are never written to memory (for the up-sampled
    first convolution of a block the bilinear x2 + modulation prologue stays its own kernel).  First-order
    differentiable; the backward recovers conv*d from `out` (hg_demod_noise_lrelu_bwd with conv = NULL)."""

    @staticmethod
    def forward(ctx, x, style, weight, nzt, wn, bn, demod, upsample, act):
        from . import conv as C
        _need_gpu(x, 'modconv_stage')
        x, style, w = _f32c(x.detach()), _f32c(style.detach()), _f32c(weight.detach())
        B, K, H, W = x.shape
        N, _, k, _ = w.shape
        s1 = style + 1.0
        if upsample:
            with on_device(x.device):
                xin = torch.empty((B, K, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
                check(lib.hg_modulate_fwd(x.data_ptr(), style.data_ptr(), xin.data_ptr(), B, K, H, W, 1, _st(x)),
                      'hg_modulate_fwd')
            iscale = None
        else:
            xin, iscale = x, s1
        Hi, Wi = xin.shape[2], xin.shape[3]
        d = None
        if demod:
            wsq = C.cached(w, 'wsq', lambda t: t.pow(2).sum(dim=(2, 3)))
            d = torch.rsqrt(_skinny_mm(s1 * s1, wsq, True) + 1e-8)
        if act:
            nzt_, wn_, bn_ = _f32c(nzt.detach()), _f32c(wn.detach().reshape(-1)), _f32c(bn.detach())
            S = nzt_.shape[-1]
        else:
            nzt_ = wn_ = bn_ = None
            S = 0
        wt = C.pack_weights(w, C.PACK_FWD)
        out = C.modconv_fwd_packed(xin, wt, N, k, iscale, d, bn_, wn_, nzt_, S, 0.2 if act else 0.0)
        ctx.save_for_backward(x, xin if upsample else None, style, w, d, out, nzt_, wn_, bn_)
        ctx.cfg = (bool(demod), bool(upsample), bool(act))
        return out

    @staticmethod
    def backward(ctx, g):
        from . import conv as C
        x, xin, style, w, d, out, nzt_, wn_, bn_ = ctx.saved_tensors
        demod, upsample, act = ctx.cfg
        g = _f32c(g.detach())
        B, K, H, W = x.shape
        N, _, k, _ = w.shape
        s1 = style + 1.0
        if xin is None:
            xin = x
        Hi, Wi = xin.shape[2], xin.shape[3]
        gwn = gbn = gd = None
        with on_device(x.device):
            if act:
                S = nzt_.shape[-1]
                gconv = torch.empty_like(out)
                gdr = torch.empty((B, N), dtype=torch.float32, device=x.device) if d is not None else None
                gw_p = torch.empty((B, N), dtype=torch.float32, device=x.device)
                gb_p = torch.empty((B, N), dtype=torch.float32, device=x.device)
                ws, n = _ws(out, B, N, Hi, Wi)
                check(lib.hg_demod_noise_lrelu_bwd(g.data_ptr(), out.data_ptr(), None,
                                                   None if d is None else d.data_ptr(), nzt_.data_ptr(), wn_.data_ptr(),
                                                   bn_.data_ptr(), gconv.data_ptr(),
                                                   None if gdr is None else gdr.data_ptr(), gw_p.data_ptr(),
                                                   gb_p.data_ptr(), B, N, Hi, S, ws.data_ptr(), n, _st(x)),
                      'hg_demod_noise_lrelu_bwd')
                gwn, gbn = gw_p.sum(0).reshape(-1, 1), gb_p.sum(0)
                gd = gdr
            else:
                if d is not None:
                    raise RuntimeError('modconv_stage: demodulation without activation is not implemented')
                gconv = g
            t = C.conv_dgrad_packed(gconv, C.pack_weights(w, C.PACK_DGRAD), K, Hi, Wi, k)
            gx = torch.empty_like(x)
            gs = torch.empty_like(style)
            ws, n = _ws(x, B, K, H, W)
            check(lib.hg_modulate_bwd(t.data_ptr(), x.data_ptr(), style.data_ptr(), gx.data_ptr(), gs.data_ptr(),
                                      B, K, H, W, int(upsample), ws.data_ptr(), n, _st(x)), 'hg_modulate_bwd')
            gw = C.conv_wgrad(xin, gconv, k, iscale=None if upsample else s1)
        if d is not None:
            wsq = w.pow(2).sum(dim=(2, 3))
            gq = gd * (-0.5) * d * d * d
            gs = gs + 2.0 * s1 * torch.mm(gq, wsq)
            gw = gw + 2.0 * w * torch.mm(gq.t(), s1 * s1)[:, :, None, None]
        return gx, gs, gw, None, gwn, gbn, None, None, None


def modconv_stage(x, style, weight, nzt=None, wn=None, bn=None, demod=True, upsample=False, act=True):
    """act(demod * conv(up?(x)*(style+1), weight) + wn*nzt + bn) -- see _ModConvStage."""
    return _ModConvStage.apply(x, style, weight, nzt, wn, bn, demod, upsample, act)


def _skinny_mm(a, m, m_transposed):
    """a (B, R) @ M (R, C) for a small B and a deep reduction R, M = m (R, C) or m.t() with m (C, R), m contiguous.
    rocBLAS has no split-K pick for these: 32 x 2048 x 2048 runs as 64 workgroups, 60 us forward / 320 us for the
    transposed operand; as S batches over chunks of R (a strided bmm, no copies) plus a sum it is S times the workgroups."""
    B, R = a.shape
    S = min(16, R // 128)
    if not SKINNY_SPLIT or S < 4 or R % S or not m.is_contiguous() or not a.is_contiguous():
        return torch.mm(a, m.t() if m_transposed else m)
    r = R // S
    av = a.view(B, S, r).transpose(0, 1)
    mv = m.view(m.shape[0], S, r).permute(1, 2, 0) if m_transposed else m.view(S, r, m.shape[1])
    return torch.bmm(av, mv).sum(0)


class _DemodCoeff(torch.autograd.Function):
    """d[b,o] = rsqrt( sum_i (y[b,i]+1)^2 * wsq[o,i] + 1e-8 ),  wsq[o,i] = sum_k W[o,i,k]^2   (Conv2DMod demodulation,
    histoGAN/histoGAN.py:427-429, on the shared weight).  wsq only depends on the weight, so it is cached per optimizer
    step for registered training weights (conv.cached) instead of re-reducing up to 151 MB three times a step."""

    @staticmethod
    def forward(ctx, y, weight):
        from . import conv as C
        w = weight.detach()
        wsq = C.cached(w, 'wsq', lambda t: t.pow(2).sum(dim=(2, 3)))
        s1 = y.detach() + 1.0
        d = torch.rsqrt(_skinny_mm(s1 * s1, wsq, True) + 1e-8)
        ctx.save_for_backward(s1, wsq, d, w)
        return d

    @staticmethod
    def backward(ctx, gd):
        s1, wsq, d, w = ctx.saved_tensors
        gy = gw = gq = None
        fused = FUSED_DEMOD_BWD and gd.is_cuda and not torch.is_grad_enabled() and wsq.is_contiguous()
        if ctx.needs_input_grad[0]:
            if fused:       # one kernel pair (hg_demod_style_grad) instead of a skinny rocBLAS GEMM + five element-wise launches
                gdc, B, N, K = _f32c(gd), d.shape[0], d.shape[1], s1.shape[1]
                with on_device(gd.device):
                    gy = torch.empty_like(s1)
                    nb = lib.hg_demod_style_grad_workspace_bytes(B, N, K)
                    ws = torch.empty(nb, dtype=torch.uint8, device=gd.device)
                    check(lib.hg_demod_style_grad(gdc.data_ptr(), d.data_ptr(), s1.data_ptr(), wsq.data_ptr(), gy.data_ptr(),
                                                  B, N, K, ws.data_ptr(), nb, _st(gd)), 'hg_demod_style_grad')
            else:
                gq = gd * (-0.5) * d * d * d
                gy = 2.0 * s1 * _skinny_mm(gq, wsq, False)
        if ctx.needs_input_grad[1]:
            from . import conv as C
            # training: added to the weight's flat gradient slot on the side stream (one kernel, hg_demod_weight_term)
            if not (w.is_contiguous() and C.direct_demod_weight_term(w, gd, d, s1)):
                gq = gd * (-0.5) * d * d * d if gq is None else gq
                gw = 2.0 * w * torch.mm(gq.t(), s1 * s1)[:, :, None, None]
        return gy, gw


def demod_coeff(y, weight):
    return _DemodCoeff.apply(y, weight)


# ---- grouped linear layers (include/hg_linear.h): the generator's 21 style projections as one launch per pass ----------
GROUPED_STYLES = os.environ.get('HG_GROUPED_STYLES', '1') != '0'   # 0: one F.linear (library GEMM) per projection


def _glin_table(xs, ws, bs, ys, groups, gws=None, gbs=None):
    from ._lib import GlinLayer
    n = len(ws)
    tab = (GlinLayer * n)()
    for i in range(n):
        t = tab[i]
        t.x, t.w, t.y = xs[groups[i]].data_ptr(), ws[i].data_ptr(), ys[i].data_ptr()
        t.b = bs[i].data_ptr() if bs is not None and bs[i] is not None else None
        t.gw = gws[i].data_ptr() if gws is not None else None
        t.gb = gbs[i].data_ptr() if gbs is not None and gbs[i] is not None else None
        t.N, t.group = ws[i].shape[0], groups[i]
    return tab


def grouped_linear_supported(xs, ws):
    B, K = xs[0].shape
    return (GROUPED_STYLES and xs[0].is_cuda and B <= 64 and K % 32 == 0 and len(ws) <= 32
            and all(w.shape[0] % 4 == 0 and w.shape[1] == K for w in ws))


class _GroupedLinear(torch.autograd.Function):
    """FIRST-ORDER ONLY (once_differentiable backward): a double backward through the generator's style projections
    (create_graph=True through G) raises -- set HG_GROUPED_STYLES=0 for such uses (one F.linear per projection,
    differentiable to any order); the trainer's path-length term is a finite difference and its gradient penalty is D-only.

    y_l = x_g(l) @ W_l^T + b_l for a list of nn.Linear layers whose inputs come in groups (histoGAN/histoGAN.py:372, 450,
    454: to_style1 / to_style2 / to_rgb.to_style of one generator block share the block's style vector).  ONE launch
    forward (hg_grouped_linear_fwd), three backward (input gradients: two, parameter gradients: one)."""

    @staticmethod
    def forward(ctx, groups, n_groups, *tensors):
        xs = [t.detach().contiguous() for t in tensors[:n_groups]]
        n = (len(tensors) - n_groups) // 2
        ws = [t.detach() for t in tensors[n_groups:n_groups + n]]
        bs = [t.detach() for t in tensors[n_groups + n:]]
        B, K = xs[0].shape
        dev = xs[0].device
        ys = [torch.empty((B, w.shape[0]), dtype=torch.float32, device=dev) for w in ws]
        tab = _glin_table(xs, ws, bs, ys, groups)
        with on_device(dev):
            check(lib.hg_grouped_linear_fwd(tab, len(ws), B, K, raw_stream(dev)), 'hg_grouped_linear_fwd')
        ctx.groups, ctx.n_groups, ctx.n = groups, n_groups, n
        ctx.save_for_backward(*xs, *ws)
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gys):
        groups, G, n = ctx.groups, ctx.n_groups, ctx.n
        saved = ctx.saved_tensors
        xs, ws = list(saved[:G]), list(saved[G:])
        B, K = xs[0].shape
        dev = xs[0].device
        gys = [(g.contiguous() if g is not None else torch.zeros((B, w.shape[0]), dtype=torch.float32, device=dev))
               for g, w in zip(gys, ws)]
        need_x = any(ctx.needs_input_grad[2:2 + G])
        need_p = any(ctx.needs_input_grad[2 + G:])
        gxs = [None] * G
        gws, gbs = [None] * n, [None] * n
        with on_device(dev):
            st = raw_stream(dev)
            if need_x:
                gxs = [torch.empty_like(x) for x in xs]
                tab = _glin_table(xs, ws, None, gys, groups)
                nb = lib.hg_grouped_linear_bwd_input_workspace_bytes(tab, n, B, K)
                wsb = torch.empty((max(nb, 4),), dtype=torch.uint8, device=dev)
                ptrs = (ctypes.c_void_p * G)(*[g.data_ptr() for g in gxs])
                check(lib.hg_grouped_linear_bwd_input(tab, n, ptrs, G, B, K, wsb.data_ptr(), wsb.numel(), st),
                      'hg_grouped_linear_bwd_input')
            if need_p:
                gws = [torch.empty_like(w) for w in ws]
                gbs = [torch.empty((w.shape[0],), dtype=torch.float32, device=dev) for w in ws]
                tab = _glin_table(xs, ws, None, gys, groups, gws, gbs)
                check(lib.hg_grouped_linear_bwd_params(tab, n, B, K, st), 'hg_grouped_linear_bwd_params')
        return (None, None, *gxs, *gws, *gbs)


def grouped_linear(xs, layers, groups):
    """xs: list of (B, K) inputs; layers: list of nn.Linear (with bias); groups[i]: index into xs of layer i's input (non-
    decreasing).  Returns the list of outputs."""
    ws = [m.weight for m in layers]
    bs = [m.bias for m in layers]
    return list(_GroupedLinear.apply(tuple(groups), len(xs), *xs, *ws, *bs))
