"""Dense contraction of the HistoGAN networks on the hand-written fp32-MFMA kernels of include/hg_conv.h.

`conv2d(x, w, bias=None, stride=1)` == `F.conv2d(x, w, bias, stride=stride, padding=k//2)` for k in {1, 3}
(stride 2 only for k = 3) -- the contraction inside Conv2DMod.forward (histoGAN/histoGAN.py:431-439, executed
with the shared weight on modulated activations) and every convolution of the discriminator (:510-518).

A convolution is bilinear in (x, w), so its output, data gradient and weight gradient are closed under
differentiation.  The three autograd Functions below call each other in their backward passes, which makes
the op differentiable to any order with only the three kernels (k_conv forward/dgrad, k_wgrad) -- the
gradient penalty (histoGAN/histoGAN.py:156-163) needs the second order.
"""
import contextlib
import ctypes
import os
import weakref

import torch

from ._lib import check, lib, on_device, raw_stream

PACK_FWD, PACK_DGRAD = 0, 1
_skip_wgrad = False

# Winograd F(2x2, 3x3) for the 3x3 stride-1 output / data-gradient launches (include/hg_wino.h): the packed operand of a
# weight carries its transformed twin as the attribute `.wino` (registered weights: written by the batched pack; others:
# packed on first use from `.wino_src`); a launch takes it when hg_wino_supported says the shape is served and faster.
WINO = os.environ.get('HG_WINO', '1') != '0'
_wino_ok = {}
_wino_used = set()    # keys of registered weights some launch took the Winograd operand of


def wino_supported(B, K, N, H, W):
    key = (B, K, N, H, W)
    r = _wino_ok.get(key)
    if r is None:
        r = _wino_ok[key] = bool(WINO and lib.hg_wino_supported(B, K, N, H, W))
    return r


_wino_wg_ok = {}
WINO_WGRAD = os.environ.get('HG_WINO_WGRAD', '1') != '0'


def wino_wgrad_supported(B, K, N, H, W):
    key = (B, K, N, H, W)
    r = _wino_wg_ok.get(key)
    if r is None:
        r = _wino_wg_ok[key] = bool(WINO and WINO_WGRAD and lib.hg_wino_wgrad_supported(B, K, N, H, W))
    return r


def _wino_pack(w, mode):
    Co, Ci = w.shape[:2]
    n = lib.hg_wino_packed_elems(Co, Ci, mode)
    if not n:
        return False
    with on_device(w.device):
        u = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib.hg_wino_pack_weights(w.data_ptr(), u.data_ptr(), Co, Ci, mode, _st(w)), 'hg_wino_pack_weights')
    return u


def _wino_u(wt, mode, B, K, N, H, W):
    """The Winograd operand for this launch, or None (direct kernel)."""
    if not wino_supported(B, K, N, H, W):
        return None
    u = getattr(wt, 'wino', None)
    src = getattr(wt, 'pack_src', None)
    if src is not None and u is not None and u is not False:
        _wino_used.add(src[2])
    if u is None:
        src = getattr(wt, 'wino_src', None)
        if src is None:
            return None
        u = wt.wino = _wino_pack(src, mode)
    return None if u is False else u


def wino_conv(x, u, N, iscale=None, oscale=None, bias=None, noise_w=None, noise_img=None, noise_S=0, slope=0.0, addend=None):
    """hg_wino_conv2d: the fused epilogue of hg_modconv2d_fwd / hg_conv2d_fwd_add on the Winograd form."""
    B, K, H, W = x.shape
    with on_device(x.device):
        out = torch.empty((B, N, H, W), dtype=torch.float32, device=x.device)
        nb = lib.hg_wino_workspace_bytes(B, K, N, H, W)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
        check(lib.hg_wino_conv2d(x.data_ptr(), u.data_ptr(), out.data_ptr(), _ptr(iscale), _ptr(oscale), _ptr(bias),
                                 _ptr(noise_w), _ptr(noise_img), noise_S, float(slope), _ptr(addend), B, K, N, H, W,
                                 _ptr(ws), nb, _st(x)), 'hg_wino_conv2d')
    return out


def modconv_fwd_packed(x, wt, N, ksize, iscale=None, oscale=None, bias=None, noise_w=None, noise_img=None, noise_S=0,
                       slope=0.0):
    """hg_modconv2d_fwd (stride 1): lrelu(oscale * conv(iscale * x) + bias + noise_w * noise_img) in one launch."""
    B, K, H, W = x.shape
    if ksize == 3 and (noise_img is None or noise_S % 2 == 0):
        u = _wino_u(wt, PACK_FWD, B, K, N, H, W)
        if u is not None:
            return wino_conv(x, u, N, iscale, oscale, bias, noise_w, noise_img, noise_S, slope)
    with on_device(x.device):
        out = torch.empty((B, N, H, W), dtype=torch.float32, device=x.device)
        nb = lib.hg_conv2d_workspace_bytes(B, K, N, H, W, ksize, 1, 0)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
        check(lib.hg_modconv2d_fwd(x.data_ptr(), _direct_operand(wt).data_ptr(), out.data_ptr(), _ptr(iscale), _ptr(oscale), _ptr(bias),
                                   _ptr(noise_w), _ptr(noise_img), noise_S, float(slope), B, K, N, H, W, ksize, _ptr(ws), nb,
                                   _st(x)), 'hg_modconv2d_fwd')
    return out


@contextlib.contextmanager
def input_grads_only():
    """Inside this context the backward of conv2d does not compute weight / bias gradients (they come back as
    None).  For `torch.autograd.grad(out, inputs=images, create_graph=True)` (the gradient penalty), where the
    autograd engine would otherwise make every conv node produce a weight gradient nobody asked for."""
    global _skip_wgrad
    old, _skip_wgrad = _skip_wgrad, True
    try:
        yield
    finally:
        _skip_wgrad = old


def _st(t):
    return raw_stream(t.device)


def _f32c(t):
    t = t.detach()
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _out_size(n, stride):
    return (n - 1) // stride + 1


def _check_args(x, w, stride):
    if not x.is_cuda:
        raise RuntimeError(f'conv2d: tensor on {x.device}; the MI355X-native path has no CPU implementation')
    Co, Ci, k, k2 = w.shape
    if k != k2 or k not in (1, 3) or stride not in (1, 2) or (stride == 2 and k != 3):
        raise ValueError(f'conv2d: weight {tuple(w.shape)} stride {stride} not supported (1x1 / 3x3 stride 1, 3x3 stride 2)')


# Packed-weight cache.  Only tensors registered with `enable_pack_cache(params)` are cached (keyed by their
# address): the Trainer registers its convolution weights, which live in one flat buffer for the life of the
# model and only change through DiffGrad.step / ema_update / load_state_dict -- the first two update through
# raw pointers (no version counter moves) and therefore call `weights_changed()`.  Everything else (plain
# torch users, the temporaries of a double backward) is packed on every call.
_cacheable = {}      # (data_ptr, shape) of a registered weight -> (owner = base address of its flat buffer, weakref to the parameter)
_cache = {}
_owner_gen = {}      # owner -> generation counter, bumped when that buffer's weights change


def _owner_of(t):
    return t.untyped_storage().data_ptr()


def enable_pack_cache(params=None):
    """Register convolution weights (4-D tensors among `params`) for packed-weight caching; None switches caching off
    and drops every registration.  Registration is ADDITIVE (a HistoGAN Trainer and a recoloring Trainer built side by
    side -- the reference CLI does that -- both keep their entries) and holds only a weak reference to the parameter:
    entries of a freed model are dropped on lookup, so a later tensor allocated at the same address cannot hit them."""
    import weakref
    if params is None:
        _cache.clear()
        _cacheable.clear()
        _multi.clear()
        return
    for p in params:
        if p.dim() == 4:
            key = (p.data_ptr(), tuple(p.shape))
            for k in [k for k in _cache if k[0] == key]:
                del _cache[k]
            _cacheable[key] = (_owner_of(p), weakref.ref(p))


def _registered_owner(w, key):
    """Owner id of a registered, still-alive weight at `key`; None (and the stale entry dropped) otherwise."""
    ent = _cacheable.get(key)
    if ent is None:
        return None
    owner, ref = ent
    p = ref()
    if p is None or p.data_ptr() != key[0]:
        del _cacheable[key]
        for k in [k for k in _cache if k[0] == key]:
            del _cache[k]
        return None
    return owner


def weights_changed(flat=None):
    """Invalidate the cached derived tensors (packed operands, squared-weight sums) of the weights living in the flat
    buffer `flat` (a tensor) -- or of all registered weights.  Called by the fused optimizer / EMA kernels, which
    update parameters through raw pointers (no version counter moves)."""
    if flat is None:
        for k in list(_owner_gen):
            _owner_gen[k] += 1
        _owner_gen[None] = _owner_gen.get(None, 0) + 1
        if _pack_events and not torch.cuda.is_current_stream_capturing():
            drain_pack_streams()
    else:
        o = _owner_of(flat)
        _owner_gen[o] = _owner_gen.get(o, 0) + 1


def _stamp(w, owner):
    return (_owner_gen.get(owner, 0), _owner_gen.get(None, 0), w._version)


def cached(w, tag, compute):
    """compute(w) cached per registered weight and `tag` until that weight's buffer changes; uncached otherwise."""
    key = (w.data_ptr(), tuple(w.shape))
    owner = _registered_owner(w, key)
    if owner is None:
        return compute(w)
    hit = _cache.get((key, tag))
    st = _stamp(w, owner)
    if hit is not None and hit[0] == st:
        if tag == 'wsq':
            _await_pack(owner, w.device, key)
        return hit[1]
    if tag == 'wsq' and PACK_MULTI and w.is_cuda and _pack_owner(owner, w.device):
        # the batched packing launch of this weight's flat buffer also leaves sum_taps W^2 of every weight (hg_pack_item.wsq)
        hit = _cache.get((key, tag))
        if hit is not None and hit[0] == st:
            return hit[1]
    val = compute(w)
    _cache[(key, tag)] = (st, val)
    return val


def pack_weights(w, mode):
    """(Co,Ci,k,k) -> the packed operand Wt[k*k][Kp][Np] of hg_conv2d_fwd / hg_conv2d_dgrad."""
    key = (w.data_ptr(), tuple(w.shape))
    owner = _registered_owner(w, key)
    if owner is not None:
        hit = _cache.get((key, mode))
        st = _stamp(w, owner)
        if hit is not None and hit[0] == st:
            _await_pack(owner, w.device, key)
            return hit[1]
        # A registered (training) weight needs both operands once per optimizer step, and so do all its siblings in the
        # same flat buffer: ONE launch packs every convolution weight of that buffer (hg_conv_pack_weights_multi) the
        # first time one of them is asked for after the buffer changed.
        if PACK_MULTI and _pack_owner(owner, w.device):
            hit = _cache.get((key, mode))
            if hit is not None and hit[0] == st:
                return hit[1]
        both = _pack_both(w)
        for m in (PACK_FWD, PACK_DGRAD):
            _cache[(key, m)] = (st, both[m])
        return both[mode]
    return _pack_weights(w, mode)


PACK_MULTI = os.environ.get('HG_PACK_MULTI', '1') != '0'
_multi = {}          # owner -> dict(sig, keys, bufs, table, n, blocks): the batched-pack plan of one flat buffer


class _PackItem(ctypes.Structure):       # include/hg_conv.h: hg_pack_item
    _fields_ = [('w', ctypes.c_void_p), ('wt_fwd', ctypes.c_void_p), ('wt_dgrad', ctypes.c_void_p),
                ('Co', ctypes.c_int32), ('Ci', ctypes.c_int32), ('ksize', ctypes.c_int32), ('block_begin', ctypes.c_int32),
                ('wsq', ctypes.c_void_p)]


class _WinoItem(ctypes.Structure):       # include/hg_wino.h: hg_wino_pack_item
    _fields_ = [('w', ctypes.c_void_p), ('u_fwd', ctypes.c_void_p), ('u_dgrad', ctypes.c_void_p), ('wsq', ctypes.c_void_p),
                ('Co', ctypes.c_int32), ('Ci', ctypes.c_int32), ('block_begin', ctypes.c_int32), ('reserved', ctypes.c_int32)]


def build_pack_plans(device):
    """Build the batched-pack plans of every registered flat buffer now (allocations + one small host-to-device copy each),
    e.g. before a hipGraph capture, inside which a plan cannot be built."""
    for owner in {own for own, _ in _cacheable.values()}:
        _pack_owner(owner, device, launch=False)


LAZY_DIRECT = os.environ.get('HG_LAZY_DIRECT', '1') != '0'   # no direct pack for weights only ever used through Winograd
_direct_needed = set()    # keys of registered 3x3 weights whose DIRECT operands some launch asked for


def _direct_operand(wt):
    """`wt` as the operand of a DIRECT launch: packed now if the batched pack skipped it (see _pack_owner)."""
    if getattr(wt, 'direct_stale', False):
        ref, mode, key = wt.pack_src
        p = ref()
        if p is None:       # (an operand outliving its weight: nothing to pack from -- never hand a stale buffer to a launch)
            raise RuntimeError('direct convolution operand requested for a weight that no longer exists')
        Co, Ci, k, _ = p.shape
        with on_device(p.device):
            check(lib.hg_conv_pack_weights(p.data_ptr(), wt.data_ptr(), Co, Ci, k, mode, _st(p)), 'hg_conv_pack_weights')
        wt.direct_stale = False
        _direct_needed.add(key)
        if p.is_cuda and not torch.cuda.is_current_stream_capturing():
            # another stream that takes this operand afterwards sees direct_stale == False: it has to wait for THIS pack
            # (registered like a group of the batched pack: consumers wait once per stream, _await_pack)
            cur = torch.cuda.current_stream(p.device)
            _pack_events.setdefault(_owner_of(p), []).append([cur.record_event(), {cur.cuda_stream}, {key}])
    return wt


PACK_SPLIT = float(os.environ.get('HG_PACK_SPLIT', '0'))   # share of a buffer's weights in the first pack group (0: one group)


def _pack_owner(owner, device, launch=True, record_on=None):
    """Pack every live registered weight of flat buffer `owner` into persistent operand buffers (direct operands + squared
    sums: hg_conv_pack_weights_multi; Winograd operands of the 3x3 weights: hg_wino_pack_weights_multi) and stamp their cache
    entries.  With HG_PACK_SPLIT > 0 the weights go in TWO groups of launches, in registration (= forward) order: the
    leading weights that together hold <= PACK_SPLIT of the bytes first (the discriminator's first five blocks are 5 % of
    its weights: its forward pass could start behind a ~20 us pack while the bulk is packed beside it).  Measured at C3:
    793.7 images/s with one group, 789.9 with the split (profiles/r05_ab_pack_split.json) -- no gain, the default is one group.  `record_on`: the stream the launches run on -- an event per group is recorded
    there for the consumers (_await_pack).  The plan (buffers + device tables) is rebuilt when the set of weights changes."""
    live = []
    for key, (own, ref) in list(_cacheable.items()):
        p = ref()
        if own == owner and p is not None and p.data_ptr() == key[0] and p.is_cuda and p.device == device \
                and p.dtype == torch.float32 and p.is_contiguous() and p.shape[2] == p.shape[3] and p.shape[2] in (1, 3):
            live.append((key, p))
    if not live:
        return False
    # A registered 3x3 weight that launches asked the DIRECT operand of and never the Winograd one (the discriminator's stride-2
    # convolutions, layers hg_wino_supported refuses) gets no Winograd operands from the next plan on: they were 16/9 of the
    # weight per mode, re-packed every optimizer step for nobody.
    no_wino = frozenset(k for k, _ in live if k in _direct_needed and k not in _wino_used)
    sig = (tuple(k for k, _ in live), frozenset(k for k, _ in live if k in _direct_needed), no_wino)
    plan = _multi.get(owner)
    with on_device(device):
        if plan is None or plan['sig'] != sig:
            if torch.cuda.is_current_stream_capturing():
                return False          # (the table upload is not capturable: per-weight launches for this capture)
            total = sum(p.numel() for _, p in live)
            ncut, run = 0, 0
            for _, p in live:
                if run + p.numel() > PACK_SPLIT * total:
                    break
                run += p.numel()
                ncut += 1
            cuts = [(0, ncut), (ncut, len(live))] if 0 < ncut < len(live) else [(0, len(live))]
            bufs, groups, lazy = {}, [], set()
            for lo, hi in cuts:
                items, witems, blocks, wblocks = [], [], 0, 0
                for key, p in live[lo:hi]:
                    Co, Ci, k, _ = p.shape
                    wf = torch.empty(lib.hg_conv_packed_elems(Co, Ci, k, PACK_FWD), dtype=torch.float32, device=device)
                    wd = torch.empty(lib.hg_conv_packed_elems(Co, Ci, k, PACK_DGRAD), dtype=torch.float32, device=device)
                    # sum over the taps of W^2 (the demodulation coefficient's weight factor), from the same tile
                    wq = torch.empty((Co, Ci), dtype=torch.float32, device=device)
                    bufs[key] = (wf, wd, wq)
                    wf.wino = wd.wino = False
                    nf = nd = 0
                    if WINO and k == 3 and key not in no_wino:   # the Winograd operands of the 3x3 weights (include/hg_wino.h), one more launch
                        nf, nd = lib.hg_wino_packed_elems(Co, Ci, PACK_FWD), lib.hg_wino_packed_elems(Co, Ci, PACK_DGRAD)
                        if nf:
                            wf.wino = torch.empty(nf, dtype=torch.float32, device=device)
                        if nd:
                            wd.wino = torch.empty(nd, dtype=torch.float32, device=device)
                    # A weight whose launches all take the Winograd operands needs no direct pack: its direct operands are
                    # left stale (`direct_stale`) until a launch asks for them -- _direct_operand packs that one weight then
                    # and puts it on the `_direct_needed` list, i.e. into the batched launch from the next plan on.
                    skip = LAZY_DIRECT and nf and nd and key not in _direct_needed
                    if skip:
                        lazy.add(key)
                    else:
                        items.append(_PackItem(p.data_ptr(), wf.data_ptr(), wd.data_ptr(), Co, Ci, k, blocks, wq.data_ptr()))
                        blocks += lib.hg_conv_pack_blocks(Co, Ci)
                    if nf or nd:
                        witems.append(_WinoItem(p.data_ptr(), wf.wino.data_ptr() if nf else None,
                                                wd.wino.data_ptr() if nd else None, wq.data_ptr() if skip else None,
                                                Co, Ci, wblocks, 0))
                        wblocks += lib.hg_wino_pack_blocks(Co, Ci, int(bool(nf)), int(bool(nd)))
                    for m_, t_ in ((PACK_FWD, wf), (PACK_DGRAD, wd)):
                        t_.pack_src = (weakref.ref(p), m_, key)
                up = lambda arr: torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(device)
                groups.append(dict(keys={key for key, _ in live[lo:hi]}, table=up((_PackItem * len(items))(*items)) if items else None,
                                   n=len(items), blocks=blocks, wn=len(witems), wblocks=wblocks,
                                   wtable=up((_WinoItem * len(witems))(*witems)) if witems else None))
            plan = _multi[owner] = dict(sig=sig, bufs=bufs, groups=groups, lazy=lazy)
        if not launch:
            return True
        _await_pack(owner, device)         # an asynchronous pack of the same buffers still in flight goes first
        _pack_events.pop(owner, None)
        events = []
        for key in plan['lazy']:           # (weights changed: a direct operand packed on demand last step is stale again)
            wf, wd, _ = plan['bufs'][key]
            wf.direct_stale = wd.direct_stale = True
        for g in plan['groups']:
            if g['table'] is not None:
                check(lib.hg_conv_pack_weights_multi(g['table'].data_ptr(), g['n'], g['blocks'], raw_stream(device)),
                      'hg_conv_pack_weights_multi')
            if g['wtable'] is not None:
                check(lib.hg_wino_pack_weights_multi(g['wtable'].data_ptr(), g['wn'], g['wblocks'], raw_stream(device)),
                      'hg_wino_pack_weights_multi')
            if record_on is not None:
                events.append([record_on.record_event(), {record_on.cuda_stream}, g['keys']])
        if events:
            _pack_events[owner] = events
    for key, p in live:
        st = _stamp(p, owner)
        wf, wd, wq = plan['bufs'][key]
        _cache[(key, PACK_FWD)] = (st, wf)
        _cache[(key, PACK_DGRAD)] = (st, wd)
        _cache[(key, 'wsq')] = (st, wq)
    return True


PREPACK = os.environ.get('HG_PREPACK', '1') != '0'
_pack_streams = {}
_pack_events = {}    # owner -> [[event behind a group of the asynchronous batched pack, ids of the streams that already wait for it, keys of the group]]


def prepack_async(flat):
    """Pack the convolution weights of flat buffer `flat` NOW, on a stream of its own, instead of at their first use: after the
    generator's optimizer step that first use is the first convolution of the next step's generator forward, behind the
    mapping network's serial ~15 us launches -- the 0.2 ms packing launch runs under those.  Consumers wait for the pack's
    event (_await_pack, once per stream); the launch itself is ordered behind everything enqueued on the current stream
    (the readers of the operand buffers it overwrites)."""
    if not (PREPACK and PACK_MULTI) or not flat.is_cuda or torch.cuda.is_current_stream_capturing():
        return False
    device, owner = flat.device, _owner_of(flat)
    st = _pack_streams.get(device.index)
    if st is None:
        st = _pack_streams[device.index] = torch.cuda.Stream(device=device)
    st.wait_event(torch.cuda.current_stream(device).record_event())
    with torch.cuda.stream(st):
        ok = _pack_owner(owner, device, record_on=st)
    return ok


def drain_pack_streams():
    """The current stream of every device waits for its asynchronous pack stream, and the per-owner pack events are dropped:
    called before a train-step graph is captured / replayed (a capturing stream must not wait on an event recorded outside the
    capture) and when every cached operand is invalidated from the host (weights_changed(None): the next pack of the same
    buffers then runs behind the one in flight)."""
    for idx, st in _pack_streams.items():
        torch.cuda.current_stream(torch.device('cuda', idx)).wait_stream(st)
    _pack_events.clear()


def _await_pack(owner, device, key=None):
    """The current stream waits for the asynchronous pack of flat buffer `owner` -- for the group that holds weight `key`
    (None: every group) -- once per stream and event."""
    if torch.cuda.is_current_stream_capturing():    # (drain_pack_streams() ran before the capture began)
        return
    ents = _pack_events.get(owner)
    if ents:
        cur = torch.cuda.current_stream(device)
        for ev, seen, keys in ents:
            if (key is None or key in keys) and cur.cuda_stream not in seen:
                cur.wait_event(ev)
                seen.add(cur.cuda_stream)


def _pack_both(w):
    Co, Ci, k, _ = w.shape
    with on_device(w.device):
        wf = torch.empty(lib.hg_conv_packed_elems(Co, Ci, k, PACK_FWD), dtype=torch.float32, device=w.device)
        wd = torch.empty(lib.hg_conv_packed_elems(Co, Ci, k, PACK_DGRAD), dtype=torch.float32, device=w.device)
        check(lib.hg_conv_pack_weights_both(w.data_ptr(), wf.data_ptr(), wd.data_ptr(), Co, Ci, k, _st(w)),
              'hg_conv_pack_weights_both')
    if k == 3:
        wf.wino_src = wd.wino_src = w      # Winograd operands: packed by the first launch that takes them (_wino_u)
    return {PACK_FWD: wf, PACK_DGRAD: wd}


def _pack_weights(w, mode):
    Co, Ci, k, _ = w.shape
    n = lib.hg_conv_packed_elems(Co, Ci, k, mode)
    if n == 0:
        raise ValueError(f'conv weights {tuple(w.shape)}: only square 1x1 / 3x3 kernels are implemented')
    with on_device(w.device):
        wt = torch.empty(n, dtype=torch.float32, device=w.device)
        check(lib.hg_conv_pack_weights(w.data_ptr(), wt.data_ptr(), Co, Ci, k, mode, _st(w)), 'hg_conv_pack_weights')
    if k == 3:
        wt.wino_src = w
    return wt


def conv_fwd_packed(x, wt, N, ksize, stride=1, iscale=None, oscale=None, bias=None):
    """out[b,n] = oscale[b,n] * sum_k conv(iscale[b,k] * x[b,k], Wt[.,k,n]) + bias[n]   (x: (B,K,H,W) contiguous)."""
    B, K, H, W = x.shape
    if ksize == 3 and stride == 1:
        u = _wino_u(wt, PACK_FWD, B, K, N, H, W)
        if u is not None:
            return wino_conv(x, u, N, iscale, oscale, bias)
    with on_device(x.device):
        out = torch.empty((B, N, _out_size(H, stride), _out_size(W, stride)), dtype=torch.float32, device=x.device)
        nb = lib.hg_conv2d_workspace_bytes(B, K, N, H, W, ksize, stride, 0)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
        check(lib.hg_conv2d_fwd(x.data_ptr(), _direct_operand(wt).data_ptr(), out.data_ptr(), _ptr(iscale), _ptr(oscale), _ptr(bias),
                                B, K, N, H, W, ksize, stride, _ptr(ws), nb, _st(x)), 'hg_conv2d_fwd')
    return out


def conv_fwd_add_packed(x, wt, N, ksize, addend, bias=None, stride=1):
    """out = (conv(x, Wt) + bias[n]) + addend   (addend (B,N,Ho,Wo) contiguous: hg_conv2d_fwd_add)."""
    B, K, H, W = x.shape
    if ksize == 3 and stride == 1 and tuple(addend.shape) == (B, N, H, W):
        u = _wino_u(wt, PACK_FWD, B, K, N, H, W)
        if u is not None:
            return wino_conv(x, u, N, bias=bias, addend=addend)
    with on_device(x.device):
        out = torch.empty((B, N, _out_size(H, stride), _out_size(W, stride)), dtype=torch.float32, device=x.device)
        if addend.shape != out.shape:
            raise ValueError(f'conv2d_add: addend {tuple(addend.shape)} does not match the output {tuple(out.shape)}')
        nb = lib.hg_conv2d_workspace_bytes(B, K, N, H, W, ksize, stride, 0)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
        check(lib.hg_conv2d_fwd_add(x.data_ptr(), _direct_operand(wt).data_ptr(), out.data_ptr(), addend.data_ptr(), _ptr(bias),
                                    B, K, N, H, W, ksize, stride, _ptr(ws), nb, _st(x)), 'hg_conv2d_fwd_add')
    return out


def lrelu_bwd_channel_sum(g, out, slope, want_sum=True):
    """(g * (out > 0 ? 1 : slope), its per-channel sum or None) in one pass (hg_lrelu_bwd_channel_sum)."""
    g, out = _f32c(g), _f32c(out)
    B, C, H, W = g.shape
    with on_device(g.device):
        gm = torch.empty_like(g)
        cs = torch.empty(C, dtype=torch.float32, device=g.device) if want_sum else None
        nb = lib.hg_nets_workspace_bytes(B, C, H, W)
        ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=g.device)
        check(lib.hg_lrelu_bwd_channel_sum(g.data_ptr(), out.data_ptr(), float(slope), gm.data_ptr(), _ptr(cs), B, C, H * W,
                                           ws.data_ptr(), ws.numel(), _st(g)), 'hg_lrelu_bwd_channel_sum')
    return gm, cs


def conv_dgrad_packed(g, wt, N, H, W, ksize, stride=1, iscale=None, oscale=None):
    """Data gradient: g (B,K,Ho,Wo) -> (B,N,H,W); wt packed with PACK_DGRAD."""
    B, K = g.shape[:2]
    if ksize == 3 and stride == 1:
        u = _wino_u(wt, PACK_DGRAD, B, K, N, H, W)
        if u is not None:
            return wino_conv(g, u, N, iscale, oscale)
    with on_device(g.device):
        gin = torch.empty((B, N, H, W), dtype=torch.float32, device=g.device)
        nb = lib.hg_conv2d_workspace_bytes(B, K, N, H, W, ksize, stride, 1)
        ws = torch.empty(nb, dtype=torch.uint8, device=g.device) if nb else None
        check(lib.hg_conv2d_dgrad(g.data_ptr(), _direct_operand(wt).data_ptr(), gin.data_ptr(), _ptr(iscale), _ptr(oscale),
                                  B, K, N, H, W, ksize, stride, _ptr(ws), nb, _st(g)), 'hg_conv2d_dgrad')
    return gin


def conv_wgrad(x, gout, ksize, stride=1, iscale=None, gscale=None, out=None):
    """gw[n,k,dy,dx] = sum_{b,y,x} gscale[b,n] gout[b,n,y,x] * iscale[b,k] x[b,k,y*s+dy-p,x*s+dx-p].
    out: optional contiguous (N,K,k,k) tensor to write (e.g. the weight's slice of a flat gradient buffer)."""
    B, K, H, W = x.shape
    N = gout.shape[1]
    if ksize == 3 and stride == 1 and iscale is None and gscale is None and wino_wgrad_supported(B, K, N, H, W):
        with on_device(x.device):     # Winograd form (include/hg_wino.h): 16 instead of 36 multiplications per tile and (n, k)
            nbytes = lib.hg_wino_wgrad_workspace_bytes(B, K, N, H, W)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            gw = out if out is not None else torch.empty((N, K, 3, 3), dtype=torch.float32, device=x.device)
            check(lib.hg_wino_wgrad(x.data_ptr(), gout.data_ptr(), gw.data_ptr(), B, K, N, H, W, ws.data_ptr(), nbytes, _st(x)),
                  'hg_wino_wgrad')
        return gw
    with on_device(x.device):
        nbytes = lib.hg_conv2d_wgrad_workspace_bytes(B, K, N, H, W, ksize, stride)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=x.device)
        gw = out if out is not None else torch.empty((N, K, ksize, ksize), dtype=torch.float32, device=x.device)
        check(lib.hg_conv2d_wgrad(x.data_ptr(), gout.data_ptr(), gw.data_ptr(), _ptr(iscale), _ptr(gscale),
                                  B, K, N, H, W, ksize, stride, ws.data_ptr(), ws.numel(), _st(x)), 'hg_conv2d_wgrad')
    return gw



# ---- weight gradients straight into the optimizer's flat gradient buffer, on a side stream ----------------------
# In a plain backward pass (no graph being built) the weight gradient of a registered convolution weight is not
# handed to autograd at all: k_wgrad writes it into the weight's slice of the flat gradient buffer (optim.FlatParams)
# on a SIDE stream.  (i) no per-parameter copy into the flat buffer afterwards; (ii) the MFMA-bound weight-gradient
# kernel runs beside the HBM-bound elementwise kernels that follow the convolution in the backward pass (LeakyReLU /
# modulation / demodulation gradients, residual adds): measured 50 % of their time hidden (tools/overlap_probe.py).
# All direct writes go through the one side stream, in order, so a second contribution to the same weight in the same
# step (gradient accumulation, the gradient penalty's second-order term, the path-length pass) is simply added there;
# FlatParams.gather() makes the main stream wait for the side stream before anything reads the buffer.
SIDE_WGRAD = os.environ.get('HG_WGRAD_STREAM', '1') != '0'
GRAPH_WGRAD_INLINE = os.environ.get('HG_GRAPH_WGRAD_INLINE', '1') != '0'
DIRECT_DEMOD = os.environ.get('HG_DIRECT_DEMOD', '1') != '0'   # demodulation's weight-gradient term into the flat slot (direct_weight_term)
_slots = {}          # (data_ptr, shape) of a registered weight -> (offset, numel, weakref to the owner FlatParams)
_side_streams = {}   # device index -> torch.cuda.Stream


def register_grad_slots(flat):
    """Called by FlatParams: convolution weights (4-d parameters) of `flat` may receive their gradient directly.
    Only a weak reference to the owner is kept (a dead owner's entries are dropped on lookup)."""
    import weakref
    ref = weakref.ref(flat)
    off = 0
    for p in flat.params:
        n = p.numel()
        if p.dim() == 4 and flat.grad is not None:
            _slots[(p.data_ptr(), tuple(p.shape))] = (off, n, ref)
        off += n


def grad_slot(w):
    """(slot view, owner FlatParams) of a registered convolution weight whose gradient may be written directly right now
    (between zero_grad() and gather(), plain backward), or None."""
    if not SIDE_WGRAD or torch.is_grad_enabled():
        return None
    ent = _slots.get((w.data_ptr(), tuple(w.shape)))
    if ent is None:
        return None
    off, n, ref = ent
    flat = ref()
    if flat is None or not getattr(flat, 'direct_ok', False):
        return None
    return flat.grad[off:off + n].view(w.shape), flat


def side_stream(device):
    st = _side_streams.get(device.index)
    if st is None:
        # HG_W_STREAM_PRIO: HIP priority of the weight-gradient stream (0 normal, -1 high), an experiment knob
        st = _side_streams[device.index] = torch.cuda.Stream(
            device=device, priority=int(os.environ.get('HG_W_STREAM_PRIO', '0')))
    return st


def _assert_single_writer(device):
    """Overwrite-vs-accumulate of a flat gradient slot is decided from the host-side set `direct_written` at ENQUEUE time:
    correct only while every direct write is enqueued on ONE stream in order (the weight-gradient stream, or the capturing
    stream with HG_GRAPH_WGRAD_INLINE).  A second writer stream would silently corrupt gradients: fail loudly instead."""
    cur = torch.cuda.current_stream(device)
    assert torch.cuda.is_current_stream_capturing() or cur.cuda_stream == side_stream(device).cuda_stream, \
        'direct gradient write outside the weight-gradient stream'


def _direct_wgrad(w, x, g, stride):
    """Weight gradient of conv(x, w) for upstream gradient g into w's flat-buffer slot; False if w has no slot (or
    a graph is being recorded): the caller then returns the gradient to autograd as usual."""
    if not SIDE_WGRAD or torch.is_grad_enabled():
        return False
    key = (w.data_ptr(), tuple(w.shape))
    ent = _slots.get(key)
    if ent is None:
        return False
    off, n, ref = ent
    flat = ref()
    if flat is None:
        del _slots[key]
        return False
    if not getattr(flat, 'direct_ok', False):       # between zero_grad() and gather() only
        return False
    slot = flat.grad[off:off + n].view(w.shape)
    xc, gc = _f32c(x), _f32c(g)
    if _batch_pieces(xc, w, stride) != 1:
        return False
    skey = slot.data_ptr()
    if GRAPH_WGRAD_INLINE and torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture every fork to the side stream becomes a cross-branch dependency edge (~100 per step):
        # measured slower than the eager side stream; the direct write (no per-parameter copy) stays, on the main branch
        if skey in flat.direct_written:
            slot.add_(conv_wgrad(xc, gc, w.shape[2], stride))
        else:
            conv_wgrad(xc, gc, w.shape[2], stride, out=slot)
            flat.direct_written.add(skey)
        return True
    main = torch.cuda.current_stream(x.device)
    side = side_stream(x.device)
    side.wait_event(main.record_event())             # g (and x) are ready on the main stream
    with torch.cuda.stream(side):
        _assert_single_writer(x.device)
        if skey in flat.direct_written:
            slot.add_(conv_wgrad(xc, gc, w.shape[2], stride))
        else:
            conv_wgrad(xc, gc, w.shape[2], stride, out=slot)
            flat.direct_written.add(skey)
    xc.record_stream(side)                           # the caching allocator must not recycle them under the kernel
    gc.record_stream(side)
    return True


def direct_demod_weight_term(w, gd, d, s1):
    """The demodulation coefficient's weight-gradient term (ops._DemodCoeff) added straight to w's flat-buffer slot on
    the weight-gradient stream by one kernel (hg_demod_weight_term: read w, read / write the slot) -- instead of a
    (N x B) @ (B x K) rocBLAS GEMM (314 us at 2048 x 2048: no library kernel for a 32-deep reduction), two element-wise
    launches over the weight, a gradient tensor for autograd and the `both` add of FlatParams.gather.  False: no slot
    (or a higher-order pass / a recording graph) -- the caller returns the term to autograd."""
    if not (SIDE_WGRAD and DIRECT_DEMOD) or torch.is_grad_enabled():
        return False
    key = (w.data_ptr(), tuple(w.shape))
    ent = _slots.get(key)
    if ent is None:
        return False
    off, n, ref = ent
    flat = ref()
    if flat is None or not getattr(flat, 'direct_ok', False):
        return False
    slot = flat.grad[off:off + n].view(w.shape)
    skey = slot.data_ptr()
    gd, d, s1 = _f32c(gd), _f32c(d), _f32c(s1)
    B, N = d.shape
    K, taps = w.shape[1], w.shape[2] * w.shape[3]

    def run():
        _assert_single_writer(w.device)
        with on_device(w.device):
            check(lib.hg_demod_weight_term(w.data_ptr(), gd.data_ptr(), d.data_ptr(), s1.data_ptr(), slot.data_ptr(), B, N, K,
                                           taps, int(skey in flat.direct_written), raw_stream(w.device)),
                  'hg_demod_weight_term')
        flat.direct_written.add(skey)

    if GRAPH_WGRAD_INLINE and torch.cuda.is_current_stream_capturing():
        run()
        return True
    main = torch.cuda.current_stream(w.device)
    side = side_stream(w.device)
    side.wait_event(main.record_event())
    with torch.cuda.stream(side):
        run()
    for t in (gd, d, s1):
        t.record_stream(side)
    return True


class _Conv(torch.autograd.Function):
    """y = conv(x, w) + bias."""

    @staticmethod
    def forward(ctx, x, w, bias, stride):
        _check_args(x, w, stride)
        if x.shape[1] != w.shape[1]:
            raise ValueError(f'conv2d: x {tuple(x.shape)} does not match w {tuple(w.shape)}')
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.has_bias = stride, bias is not None
        xc, wc = _f32c(x), _f32c(w)
        bc = None if bias is None else _f32c(bias)
        return conv_fwd_packed(xc, pack_weights(wc, PACK_FWD), w.shape[0], w.shape[2], stride, bias=bc)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _ConvDgrad.apply(g, w, x.shape[2], x.shape[3], ctx.stride)
        if not _skip_wgrad:
            if ctx.needs_input_grad[1] and not _direct_wgrad(w, x, g, ctx.stride):
                gw = _ConvWgrad.apply(g, x, w.shape[2], ctx.stride)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if torch.is_grad_enabled():     # higher-order pass: keep the sum on the autograd tape
                    gb = g.sum(dim=(0, 2, 3))
                else:
                    from .ops import channel_sum
                    gb = channel_sum(g)
        return gx, gw, gb, None


class _ConvDgrad(torch.autograd.Function):
    """gx = conv^T(g, w): the data gradient of _Conv; bilinear in (g, w)."""

    @staticmethod
    def forward(ctx, g, w, H, W, stride):
        _check_args(g, w, stride)
        ctx.save_for_backward(g, w)
        ctx.stride = stride
        gc, wc = _f32c(g), _f32c(w)
        return conv_dgrad_packed(gc, pack_weights(wc, PACK_DGRAD), w.shape[1], H, W, w.shape[2], stride)

    @staticmethod
    def backward(ctx, ggx):
        g, w = ctx.saved_tensors
        gg = gw = None
        if ctx.needs_input_grad[0]:
            gg = _Conv.apply(ggx, w, None, ctx.stride)
        if ctx.needs_input_grad[1] and not _skip_wgrad and not _direct_wgrad(w, ggx, g, ctx.stride):
            gw = _ConvWgrad.apply(g, ggx, w.shape[2], ctx.stride)
        return gg, gw, None, None, None


class _ConvWgrad(torch.autograd.Function):
    """gw = sum_pixels g (x) x: the weight gradient of _Conv; bilinear in (g, x)."""

    @staticmethod
    def forward(ctx, g, x, ksize, stride):
        if not x.is_cuda:
            raise RuntimeError('conv2d: no CPU implementation')
        ctx.save_for_backward(g, x)
        ctx.stride = stride
        return conv_wgrad(_f32c(x), _f32c(g), ksize, stride)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        gg = gx = None
        if ctx.needs_input_grad[0]:
            gg = _Conv.apply(x, ggw, None, ctx.stride)
        if ctx.needs_input_grad[1]:
            gx = _ConvDgrad.apply(g, ggw, x.shape[2], x.shape[3], ctx.stride)
        return gg, gx, None, None


class _ConvLrelu(torch.autograd.Function):
    """y = leaky_relu(conv(x, w) + bias, slope) in ONE launch (the epilogue of the fused-extras k_conv instantiation):
    the `nn.Conv2d -> LeakyReLU(0.2)` pairs of DiscriminatorBlock.net (histoGAN/histoGAN.py:510-515).  The backward
    masks the incoming gradient with aten's leaky_relu_backward on the OUTPUT (sign(y) == sign(pre-activation)), which
    autograd can differentiate again, then continues with the bilinear conv Functions -> any-order differentiable."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, slope):
        _check_args(x, w, stride)
        if x.shape[1] != w.shape[1]:
            raise ValueError(f'conv2d: x {tuple(x.shape)} does not match w {tuple(w.shape)}')
        xc, wc = _f32c(x), _f32c(w)
        B, K, H, W = xc.shape
        N, k = w.shape[0], w.shape[2]
        wt = pack_weights(wc, PACK_FWD)
        bc = None if bias is None else _f32c(bias)
        if stride != 1:
            raise ValueError('conv2d_lrelu: stride 1 only')
        out = modconv_fwd_packed(xc, wt, N, k, bias=bc, slope=slope)
        ctx.save_for_backward(x, w, out)
        ctx.stride, ctx.has_bias, ctx.slope = stride, bias is not None, float(slope)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        gx = gw = gb = None
        want_gb = ctx.has_bias and ctx.needs_input_grad[2] and not _skip_wgrad
        if torch.is_grad_enabled():      # a graph of this backward is being recorded (gradient penalty): differentiable ops
            gm = torch.ops.aten.leaky_relu_backward(g, out, ctx.slope, True)
            if want_gb:
                gb = gm.sum(dim=(0, 2, 3))
        else:                            # plain backward: the mask and the bias gradient from ONE pass over (g, out)
            gm, gb = lrelu_bwd_channel_sum(g, out, ctx.slope, want_gb)
        if ctx.needs_input_grad[0]:
            gx = _ConvDgrad.apply(gm, w, x.shape[2], x.shape[3], ctx.stride)
        if not _skip_wgrad:
            if ctx.needs_input_grad[1] and not _direct_wgrad(w, x, gm, ctx.stride):
                gw = _ConvWgrad.apply(gm, x, w.shape[2], ctx.stride)
        return gx, gw, gb, None, None


class _ConvAdd(torch.autograd.Function):
    """y = (conv(x, w) + bias) + addend in ONE launch: the residual sum of DiscriminatorBlock.forward
    (`x = self.net(x); x = x + res`, histoGAN/histoGAN.py:520-524) in the epilogue of the 1x1 `conv_res` launch.
    Bilinear part as _Conv (differentiable to any order), the addend's gradient is the incoming one."""

    @staticmethod
    def forward(ctx, x, w, bias, addend, stride):
        _check_args(x, w, stride)
        if x.shape[1] != w.shape[1]:
            raise ValueError(f'conv2d: x {tuple(x.shape)} does not match w {tuple(w.shape)}')
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.has_bias = stride, bias is not None
        xc, wc = _f32c(x), _f32c(w)
        bc = None if bias is None else _f32c(bias)
        return conv_fwd_add_packed(xc, pack_weights(wc, PACK_FWD), w.shape[0], w.shape[2], _f32c(addend), bc, stride)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _ConvDgrad.apply(g, w, x.shape[2], x.shape[3], ctx.stride)
        if not _skip_wgrad:
            if ctx.needs_input_grad[1] and not _direct_wgrad(w, x, g, ctx.stride):
                gw = _ConvWgrad.apply(g, x, w.shape[2], ctx.stride)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if torch.is_grad_enabled():
                    gb = g.sum(dim=(0, 2, 3))
                else:
                    from .ops import channel_sum
                    gb = channel_sum(g)
        return gx, gw, gb, (g if ctx.needs_input_grad[3] else None), None


_I32 = 2 ** 31 - 1


def _batch_pieces(x, w, stride):
    """The kernels index activations with 32-bit element offsets: a batch whose input or output tensor has >= 2^31
    elements (the 1024^2 configuration with attention: 16 x 512 x 512 x 512) is processed in batch slices (samples
    are independent; autograd sums the weight gradients of the slices)."""
    B, K, H, W = x.shape
    per = max(K * H * W, w.shape[0] * _out_size(H, stride) * _out_size(W, stride))
    if B * per <= _I32:
        return 1
    if per > _I32:
        raise ValueError(f'conv2d: one sample of shape {tuple(x.shape[1:])} -> {w.shape[0]} channels exceeds 2^31 elements')
    return -(-B // (_I32 // per))


def _sliced(fn, x, w, stride):
    n = _batch_pieces(x, w, stride)
    if n == 1:
        return fn(x)
    return torch.cat([fn(xs) for xs in x.chunk(n, dim=0)], dim=0)


def conv2d_lrelu(x, w, bias=None, slope=0.2):
    """leaky_relu(F.conv2d(x, w, bias, padding=k//2), slope) as one launch (stride 1)."""
    return _sliced(lambda t: _ConvLrelu.apply(t, w, bias, 1, slope), x, w, 1)


def conv2d(x, w, bias=None, stride=1):
    """F.conv2d(x, w, bias, stride=stride, padding=k//2) for k in {1,3} on the MFMA implicit-GEMM kernels."""
    return _sliced(lambda t: _Conv.apply(t, w, bias, stride), x, w, stride)


def conv2d_add(x, w, bias, addend, stride=1):
    """F.conv2d(x, w, bias, stride, padding=k//2) + addend as one launch."""
    if _batch_pieces(x, w, stride) != 1:
        return conv2d(x, w, bias, stride) + addend
    return _ConvAdd.apply(x, w, bias, addend, stride)


def conv2d_same(x, w, bias=None):
    return _sliced(lambda t: _Conv.apply(t, w, bias, 1), x, w, 1)
