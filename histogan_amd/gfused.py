"""The generator's TRAINING pass as one hand-scheduled autograd node (first order).

`Generator.forward` (histoGAN/histoGAN.py:558-568) over `GeneratorBlock.forward` (:461-479), `Conv2DMod` (:420-440) and
`RGBBlock` (:380-390), with the per-block launch sequence of nets.GeneratorBlock._stage, but ONE autograd Function for the whole
network instead of ~90 nodes: the backward is written out by hand, and everything that sits between two convolutions of it
-- modulation adjoint (+ bilinear x2 adjoint), to-RGB adjoint, the sum of the two gradients of a block output, LeakyReLU /
noise / demodulation adjoint -- is ONE launch (`hg_gstage_bwd`, include/hg_nets.h) where autograd ran five to six in a row on
the critical path between one data gradient and the next (un-profiled phase probe, profiles/r06_phase_probe.txt: the
G-phase backward was 13.95 ms of a 35.2 ms step against 9.9 ms of convolution kernels in it).

The 21 style projections stay outside (ops.grouped_linear: their own node); the 14 demodulation coefficients are computed
and differentiated inside (`_demod` / `_demod_bwd`: the weight term is added to the flat gradient slot on the weight-gradient
stream right behind the convolution's weight gradient that wrote it), so that when the node's backward returns every
convolution weight of the generator has its final gradient.  HG_GFUSED=0 keeps the per-block autograd path.
"""
import os

import torch

from . import conv as C
from . import ops
from ._lib import check, lib, on_device, raw_stream

GFUSED = os.environ.get('HG_GFUSED', '1') != '0'
PER_BLOCK = 10        # tensors per block in the Function's argument list (see generator_train)
DEMOD_AUX = os.environ.get('HG_DEMOD_AUX', '0') != '0'
AFTER_BLOCKS = None    # trainer: called when the node's backward has enqueued the last convolution weight gradient
STAGE_OBSERVER = None  # tests: called with every stage output (two per block, forward order) -- the LeakyReLU branches taken


def _st(t):
    return raw_stream(t.device)


def _f32c(t):
    t = t.detach()
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def gstage_bwd(out, ga, sa, up, g_rgb, w_rgb, s_rgb, d, nzt, wn, bn, gw_rgb_out=None):
    """hg_gstage_bwd: -> (gconv, gs_a, gs_rgb, gw_rgb, gd, gwn, gbn); see include/hg_nets.h.  All tensors fp32 contiguous.
    gw_rgb_out: where to write the to-RGB weight gradient (e.g. the weight's flat gradient slot)."""
    B, Cc, H, _ = out.shape
    S = nzt.shape[-1]
    dev = out.device
    Cr = 0 if g_rgb is None else g_rgb.shape[1]
    with on_device(dev):
        gconv = torch.empty_like(out)
        gs_a = torch.empty((B, Cc), dtype=torch.float32, device=dev) if (ga is not None and sa is not None) else None
        gs_rgb = torch.empty((B, Cc), dtype=torch.float32, device=dev) if (g_rgb is not None and s_rgb is not None) else None
        gw_rgb = None
        if g_rgb is not None:
            gw_rgb = gw_rgb_out if gw_rgb_out is not None else torch.empty((Cr, Cc), dtype=torch.float32, device=dev)
        gd = torch.empty((B, Cc), dtype=torch.float32, device=dev) if d is not None else None
        gwn = torch.empty((Cc,), dtype=torch.float32, device=dev)
        gbn = torch.empty((Cc,), dtype=torch.float32, device=dev)
        nb = lib.hg_gstage_bwd_workspace_bytes(B, Cc, H, int(bool(up)))
        ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
        check(lib.hg_gstage_bwd(out.data_ptr(), _ptr(ga), _ptr(sa), int(bool(up)), _ptr(g_rgb), _ptr(w_rgb), _ptr(s_rgb), Cr,
                                _ptr(d), nzt.data_ptr(), wn.data_ptr(), bn.data_ptr(), S, gconv.data_ptr(), _ptr(gs_a),
                                _ptr(gs_rgb), _ptr(gw_rgb), _ptr(gd), gwn.data_ptr(), gbn.data_ptr(), B, Cc, H, ws.data_ptr(), nb,
                                _st(out)), 'hg_gstage_bwd')
    return gconv, gs_a, gs_rgb, gw_rgb, gd, gwn, gbn


def _modulate(x, s, upsample):
    B, Cc, H, W = x.shape
    f = 2 if upsample else 1
    with on_device(x.device):
        out = torch.empty((B, Cc, H * f, W * f), dtype=torch.float32, device=x.device)
        check(lib.hg_modulate_fwd(x.data_ptr(), _ptr(s), out.data_ptr(), B, Cc, H, W, int(upsample), _st(x)), 'hg_modulate_fwd')
    return out


def _modulate_bwd(g, x, s, upsample):
    B, Cc, H, W = x.shape
    with on_device(x.device):
        gx = torch.empty_like(x)
        gs = None if s is None else torch.empty_like(s)
        n = lib.hg_nets_workspace_bytes(B, Cc, H, W)
        ws = torch.empty(max(n, 4), dtype=torch.uint8, device=x.device)
        check(lib.hg_modulate_bwd(g.data_ptr(), x.data_ptr(), _ptr(s), gx.data_ptr(), _ptr(gs), B, Cc, H, W, int(upsample),
                                  ws.data_ptr(), n, _st(x)), 'hg_modulate_bwd')
    return gx, gs


def _torgb(x, s, w, prev):
    B, O, H, W = x.shape
    Cr = w.shape[0]
    if Cr * O * 4 > 48 * 1024:
        # (the 8 192-channel blocks of the 1024^2 configuration: hg_torgb_fwd stages w (s + 1) in LDS; there the 1x1 modulated
        # convolution runs on the matrix kernel with the modulation as its input scale, plus the running-image add)
        rgb = C.conv_fwd_packed(x, C.pack_weights(w.reshape(Cr, O, 1, 1), C.PACK_FWD), Cr, 1, 1, iscale=s + 1.0)
        return rgb if prev is None else rgb.add_(prev)
    with on_device(x.device):
        out = torch.empty((B, Cr, H, W), dtype=torch.float32, device=x.device)
        check(lib.hg_torgb_fwd(x.data_ptr(), s.data_ptr(), w.data_ptr(), _ptr(prev), out.data_ptr(), B, O, Cr, H * W, _st(x)),
              'hg_torgb_fwd')
    return out


def _wgrad(w, x, g):
    """Weight gradient of conv(x, w): into w's flat gradient slot on the weight-gradient stream (None returned), or a tensor."""
    if C._skip_wgrad or C._direct_wgrad(w, x, g, 1):
        return None
    return C.conv_wgrad(x, g, w.shape[2])


def _demod(s, w):
    """d[b,o] = rsqrt(sum_i (s[b,i]+1)^2 wsq[o,i] + 1e-8), wsq = sum_taps W^2 (Conv2DMod demodulation, histoGAN/histoGAN.py:427-429
    on the shared weight; ops._DemodCoeff without the autograd node) -> (d, s + 1, wsq)."""
    wsq = C.cached(w, 'wsq', lambda t: t.pow(2).sum(dim=(2, 3)))
    s1 = s + 1.0
    return torch.rsqrt(ops._skinny_mm(s1 * s1, wsq, True) + 1e-8), s1, wsq


def _demod_bwd(gd, d, s1, wsq, wp):
    """d's adjoint: the style part (returned, (B,K)) and the weight part -- added to the weight's flat gradient slot on the
    weight-gradient stream, BEHIND the convolution's weight gradient that wrote the slot (returns None), or returned."""
    B, N, K = d.shape[0], d.shape[1], s1.shape[1]
    gw = None
    if wsq.is_contiguous():
        with on_device(gd.device):
            gy = torch.empty_like(s1)
            nb = lib.hg_demod_style_grad_workspace_bytes(B, N, K)
            ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=gd.device)
            check(lib.hg_demod_style_grad(gd.data_ptr(), d.data_ptr(), s1.data_ptr(), wsq.data_ptr(), gy.data_ptr(), B, N, K,
                                          ws.data_ptr(), nb, _st(gd)), 'hg_demod_style_grad')
        gq = None
    else:
        gq = gd * (-0.5) * d * d * d
        gy = 2.0 * s1 * ops._skinny_mm(gq, wsq, False)
    if not (wp.is_contiguous() and C.direct_demod_weight_term(wp, gd, d, s1)):
        gq = gd * (-0.5) * d * d * d if gq is None else gq
        gw = 2.0 * wp.detach() * torch.mm(gq.t(), s1 * s1)[:, :, None, None]
    return gy, gw


def _add(a, b):
    return b if a is None else (a if b is None else a + b)


class _GeneratorTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, nzt, *ts):
        L = len(ts) // PER_BLOCK
        nzt_ = _f32c(nzt)
        S = nzt_.shape[-1]
        B = ts[0].shape[0]
        x = _f32c(x0).expand(B, -1, -1, -1).contiguous()
        saved = [x, nzt_]
        prev = None
        rgb = None
        for i in range(L):
            s1, s2, srgb, w1, w2, wrgb, wn1, bn1, wn2, bn2 = [_f32c(t) for t in ts[PER_BLOCK * i:PER_BLOCK * (i + 1)]]
            wn1, wn2 = wn1.reshape(-1), wn2.reshape(-1)
            N = w1.shape[0]
            d1, s1p, wsq1 = _demod(s1, w1)
            xm1 = _modulate(x, s1, i != 0)
            out1 = C.modconv_fwd_packed(xm1, C.pack_weights(w1, C.PACK_FWD), N, 3, None, d1, bn1, wn1, nzt_, S, 0.2)
            d2, s2p, wsq2 = _demod(s2, w2)
            xm2 = _modulate(out1, s2, False)
            out2 = C.modconv_fwd_packed(xm2, C.pack_weights(w2, C.PACK_FWD), N, 3, None, d2, bn2, wn2, nzt_, S, 0.2)
            rgb = _torgb(out2, srgb, wrgb.reshape(wrgb.shape[0], -1), prev)
            if i != L - 1:
                prev = _modulate(rgb, None, True)
            x = out2
            if STAGE_OBSERVER is not None:
                STAGE_OBSERVER(out1)
                STAGE_OBSERVER(out2)
            saved += [xm1, out1, xm2, out2, s1, s2, srgb, d1, d2, wn1, bn1, wn2, bn2, s1p, s2p, wsq1, wsq2]
        ctx.save_for_backward(*saved)
        ctx.weights = [ts[PER_BLOCK * i + 3:PER_BLOCK * i + 6] for i in range(L)]      # the PARAMETERS (flat-slot lookup by address)
        ctx.L = L
        return rgb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        L = ctx.L
        sv = ctx.saved_tensors
        x0e, nzt = sv[0], sv[1]
        per = 17
        blk = lambda i: sv[2 + per * i:2 + per * (i + 1)]
        grads = [None] * (PER_BLOCK * L)
        from . import nets as _N
        if _N.PHASE_HOOK is not None:
            _N.PHASE_HOOK('gb_generator_node_entered', True)
        g_rgb = _f32c(g)
        ga = None          # d loss / d (modulated, up-sampled input of the NEXT block's first convolution)
        sa = None
        # The demodulation coefficients' adjoints (a small GEMM-like kernel pair + the weight term per convolution) only feed
        # the style gradients returned at the end.  HG_DEMOD_AUX=1 runs them on the auxiliary stream, off the dgrad -> stage ->
        # dgrad chain: measured 854.6 vs 867.7 images/s inline (they already hide under the weight-gradient kernels) -- off.
        dev = g_rgb.device
        main, aux = torch.cuda.current_stream(dev), _N.aux_stream(dev)
        pend = []          # (index into grads, modulation part, demodulation part)
        wterm = {}

        def demod_async(idx, gd, d, s1p, wsq, wp):
            if not DEMOD_AUX or torch.cuda.is_current_stream_capturing():
                gy, wterm[idx] = _demod_bwd(gd, d, s1p, wsq, wp)
                return gy
            aux.wait_event(main.record_event())
            with torch.cuda.stream(aux):
                gy, gwd = _demod_bwd(gd, d, s1p, wsq, wp)
            gd.record_stream(aux)
            gy.record_stream(main)
            if gwd is not None:
                gwd.record_stream(main)
            wterm[idx] = gwd
            return gy
        for i in range(L - 1, -1, -1):
            xm1, out1, xm2, out2, s1, s2, srgb, d1, d2, wn1, bn1, wn2, bn2, s1p, s2p, wsq1, wsq2 = blk(i)
            w1p, w2p, wrgbp = ctx.weights[i]
            w1, w2, wrgb = _f32c(w1p), _f32c(w2p), _f32c(wrgbp)
            Cr = wrgb.shape[0]
            base = PER_BLOCK * i
            # ---- at out2: next block's first convolution (behind the bilinear x2) + this block's to-RGB -> conv2's upstream gradient
            # (the to-RGB weight gradient straight into its flat slot when that slot has no writer yet this step)
            slot = C.grad_slot(wrgbp) if wrgbp.is_contiguous() else None
            if slot is not None and slot[0].data_ptr() in slot[1].direct_written:
                slot = None
            gconv2, gs_a, gs_rgb, gw_rgb, gd2, gwn2, gbn2 = gstage_bwd(out2, ga, sa, ga is not None, g_rgb, wrgb.reshape(Cr, -1),
                                                                       srgb, d2, nzt, wn2, bn2,
                                                                       None if slot is None else slot[0].view(Cr, -1))
            if ga is not None:
                pend.append((base + PER_BLOCK + 0, gs_a, gy_next))   # style of the next block's conv1: modulation + demodulation parts
            grads[base + 2] = gs_rgb
            if slot is None:
                grads[base + 5] = gw_rgb.reshape(wrgbp.shape)
            else:
                slot[1].direct_written.add(slot[0].data_ptr())
            grads[base + 8], grads[base + 9] = gwn2.reshape(-1, 1), gbn2
            g_xm2 = C.conv_dgrad_packed(gconv2, C.pack_weights(w2, C.PACK_DGRAD), w2.shape[1], xm2.shape[2], xm2.shape[3], 3)
            grads[base + 4] = _wgrad(w2p, xm2, gconv2)
            gy2 = demod_async(base + 4, gd2, d2, s2p, wsq2, w2p)
            if i > 0:                                            # rgb_i = to_rgb(out2) + up2(rgb_{i-1})
                g_rgb_prev, _ = _modulate_bwd(g_rgb, torch.empty((g_rgb.shape[0], Cr, g_rgb.shape[2] // 2, g_rgb.shape[3] // 2),
                                                                 dtype=torch.float32, device=g_rgb.device), None, True)
            # ---- at out1: conv2 (same resolution) -> conv1's upstream gradient
            gconv1, gs2, _, _, gd1, gwn1, gbn1 = gstage_bwd(out1, g_xm2, s2, False, None, None, None, d1, nzt, wn1, bn1)
            pend.append((base + 1, gs2, gy2))
            grads[base + 6], grads[base + 7] = gwn1.reshape(-1, 1), gbn1
            ga = C.conv_dgrad_packed(gconv1, C.pack_weights(w1, C.PACK_DGRAD), w1.shape[1], xm1.shape[2], xm1.shape[3], 3)
            grads[base + 3] = _wgrad(w1p, xm1, gconv1)
            gy_next = demod_async(base + 3, gd1, d1, s1p, wsq1, w1p)
            sa = s1
            if i > 0:
                g_rgb = g_rgb_prev
        # block 0's first convolution reads the learned constant directly (no upsample)
        gx0e, gs1_0 = _modulate_bwd(ga, x0e, sa, False)
        pend.append((0, gs1_0, gy_next))
        g_x0 = gx0e.sum(0)
        if DEMOD_AUX and not torch.cuda.is_current_stream_capturing():
            main.wait_stream(aux)
        torch._foreach_add_([a for _, a, _ in pend], [b for _, _, b in pend])
        for idx, a, _ in pend:
            grads[idx] = a
        for idx, gwd in wterm.items():
            grads[idx] = _add(grads[idx], gwd)
        if _N.PHASE_HOOK is not None:
            _N.PHASE_HOOK('gb_generator_blocks_done', True)
        if AFTER_BLOCKS is not None:
            AFTER_BLOCKS()
        return (g_x0, None, *grads)


def generator_infer(gen, styles_t, nzt):
    """The same launch sequence without autograd (the D phase's generator forward, evaluate()): modulation inside the
    convolution kernel where there is no upsample in front of it, and the 14 demodulation coefficients -- six small dependent
    launches each, ~0.4 ms as links of the convolution chain -- computed AHEAD on the auxiliary stream (they depend on the
    styles and the weights only); the chain waits for its block's event."""
    dev = nzt.device
    main, aux = torch.cuda.current_stream(dev), nets_aux(dev)
    L = len(gen.blocks)
    B = styles_t[0].shape[0]
    S = nzt.shape[-1]
    aux.wait_event(main.record_event())
    ahead = []
    with torch.cuda.stream(aux):
        for i, b in enumerate(gen.blocks):
            w1, w2 = _f32c(b.conv1.weight), _f32c(b.conv2.weight)
            d1, s1p, _ = _demod(_f32c(styles_t[3 * i]), w1)
            d2, s2p, _ = _demod(_f32c(styles_t[3 * i + 1]), w2)
            ahead.append((d1, s1p, d2, s2p, aux.record_event()))
    for t in styles_t:
        t.record_stream(aux)
    x = _f32c(gen.initial_block).expand(B, -1, -1, -1).contiguous()
    prev = rgb = None
    for i, b in enumerate(gen.blocks):
        d1, s1p, d2, s2p, ev = ahead[i]
        main.wait_event(ev)
        for t in (d1, s1p, d2, s2p):
            t.record_stream(main)
        w1, w2, wrgb = _f32c(b.conv1.weight), _f32c(b.conv2.weight), _f32c(b.to_rgb.conv.weight)
        wn1, bn1 = _f32c(b.to_noise1.weight).reshape(-1), _f32c(b.to_noise1.bias)
        wn2, bn2 = _f32c(b.to_noise2.weight).reshape(-1), _f32c(b.to_noise2.bias)
        N = w1.shape[0]
        if i:
            x, isc = _modulate(x, _f32c(styles_t[3 * i]), True), None
        else:
            isc = s1p
        x = C.modconv_fwd_packed(x, C.pack_weights(w1, C.PACK_FWD), N, 3, isc, d1, bn1, wn1, nzt, S, 0.2)
        x = C.modconv_fwd_packed(x, C.pack_weights(w2, C.PACK_FWD), N, 3, s2p, d2, bn2, wn2, nzt, S, 0.2)
        rgb = _torgb(x, _f32c(styles_t[3 * i + 2]), wrgb.reshape(wrgb.shape[0], -1), prev)
        if i != L - 1:
            prev = _modulate(rgb, None, True)
    return rgb


def nets_aux(dev):
    from .nets import aux_stream
    return aux_stream(dev)


def supported(gen, styles_t, nzt, train=True):
    """Shapes / options the fused node serves (everything HistoGAN trains with); anything else takes the per-block path."""
    if not (GFUSED and torch.is_grad_enabled() == train and nzt.is_cuda and nzt.dtype == torch.float32):
        return False
    if not train and torch.cuda.is_current_stream_capturing():
        return False
    S = nzt.shape[-1]
    if S % 4 or gen.initial_block.shape[-1] % 4 or gen.initial_block.shape[-1] != gen.initial_block.shape[-2]:
        return False
    B = styles_t[0].shape[0]
    H = gen.initial_block.shape[-1]
    for i, b in enumerate(gen.blocks):
        if i:
            H *= 2
        for cv in (b.conv1, b.conv2):
            if not (cv.demod and cv.kernel == 3 and cv.stride == 1 and cv.dilation == 1 and cv.weight.dtype == torch.float32):
                return False
        r = b.to_rgb.conv
        if r.demod or r.kernel != 1 or r.stride != 1 or r.dilation != 1 or r.weight.shape[0] > 4:
            return False
        Cmax = max(b.conv1.weight.shape[0], b.conv1.weight.shape[1])
        if H > S or B * Cmax * H * H * 4 >= 2 ** 31:
            return False
        if (b.upsample is not None) != (i != 0) or (b.to_rgb.upsample is not None) != (i != len(gen.blocks) - 1):
            return False
    return all(t.dtype == torch.float32 for t in styles_t)


def generator_train(gen, styles_t, nzt):
    """rgb = Generator(...) given the 3 L projected styles `styles_t` ([to_style1, to_style2, to_rgb.to_style] per block) and the
    transposed noise image; differentiable (first order) w.r.t. the styles and every generator parameter."""
    args = []
    for i, b in enumerate(gen.blocks):
        s1, s2, srgb = styles_t[3 * i], styles_t[3 * i + 1], styles_t[3 * i + 2]
        args += [s1, s2, srgb, b.conv1.weight, b.conv2.weight, b.to_rgb.conv.weight,
                 b.to_noise1.weight, b.to_noise1.bias, b.to_noise2.weight, b.to_noise2.bias]
    return _GeneratorTrain.apply(gen.initial_block, nzt, *args)
