"""RGB-uv histogram + Hellinger loss as torch.autograd Functions over the HIP C ABI.

Host-side mirror of histogram_classes/RGBuvHistBlock.py:75-228 (forward) and of the autograd
replay of it; Hellinger loss of histoGAN/histoGAN.py:957-960.  PyTorch is used only for device
memory (caching allocator), the current stream and autograd plumbing.
"""
import os
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import HgHistParams, check, lib, on_device, raw_stream

_IDX_CACHE = {}
PROJ_CACHE = os.environ.get('HG_PROJ_CACHE', '1') != '0'   # A/B switch of the forward->backward projection cache


def _sampling_idx(size, h, device):
    """torch.LongTensor(np.linspace(0, size, h, endpoint=False)) of RGBuvHistBlock.py:82-87."""
    key = (size, h, str(device))
    t = _IDX_CACHE.get(key)
    if t is None:
        t = torch.from_numpy(np.linspace(0, size, h, endpoint=False).astype(np.int64).astype(np.int32)).to(device)
        _IDX_CACHE[key] = t
    return t


class HistConfig:
    """Immutable description of one RGBuvHistBlock (ctor args, RGBuvHistBlock.py:29-73)."""
    __slots__ = ('h', 'insz', 'resizing', 'method', 'sigma', 'intensity_scale', 'lo', 'hi', 'green_only', 'projection')

    def __init__(self, h=64, insz=150, resizing='interpolation', method='inverse-quadratic', sigma=0.02,
                 intensity_scale=True, hist_boundary=None, green_only=False, projection='rgbuv'):
        """projection: 'rgbuv' (RGBuvHistBlock, 3 planes, default boundary [-3,3]), 'rgchroma' (rgChromaHistBlock) or
        'direct' (LabHistBlock): one plane, default boundary [0,1]."""
        if projection not in _lib.HG_PROJ:
            raise ValueError(f'unknown projection {projection!r}')
        self.projection = projection
        if hist_boundary is None:
            hist_boundary = [-3, 3] if projection == 'rgbuv' else [0, 1]
        hb = sorted(hist_boundary)
        self.h, self.insz, self.resizing, self.method = int(h), insz, resizing, method
        self.sigma = sigma
        self.intensity_scale, self.green_only = bool(intensity_scale), bool(green_only)
        self.lo, self.hi = float(hb[0]), float(hb[1])


def _make_params(x, cfg, pre_relu=False):
    if x.dim() != 4 or x.shape[1] < 3:
        raise ValueError(f'expected (B, C>=3, H, W) input, got {tuple(x.shape)}')
    if cfg.method not in _lib.HG_METHOD:
        raise Exception(f'Wrong kernel method. It should be either thresholding, RBF,'
                        f' inverse-quadratic. But the given value is {cfg.method}.')
    B, C, H, W = x.shape
    p = HgHistParams()
    p.struct_size = ctypes.sizeof(HgHistParams)
    p.B, p.C, p.H, p.W = B, C, H, W
    p.stride_b, p.stride_c, p.stride_h, p.stride_w = x.stride()
    keep = []
    if H > cfg.insz or W > cfg.insz:
        if cfg.resizing == 'interpolation':
            p.resize_mode, p.Hs, p.Ws = _lib.HG_RESIZE_BILINEAR, int(cfg.insz), int(cfg.insz)
        elif cfg.resizing == 'sampling':
            r, c = _sampling_idx(H, cfg.h, x.device), _sampling_idx(W, cfg.h, x.device)
            keep = [r, c]
            p.resize_mode, p.Hs, p.Ws = _lib.HG_RESIZE_SAMPLING, cfg.h, cfg.h
            p.row_idx, p.col_idx = r.data_ptr(), c.data_ptr()
        else:
            raise Exception(f'Wrong resizing method. It should be: interpolation or sampling. '
                            f'But the given value is {cfg.resizing}.')
    else:
        p.resize_mode, p.Hs, p.Ws = _lib.HG_RESIZE_NONE, H, W
    p.h, p.lo, p.hi = cfg.h, cfg.lo, cfg.hi
    p.method = _lib.HG_METHOD[cfg.method]
    p.sigma = float(cfg.sigma) if cfg.method != 'thresholding' else 1.0
    p.intensity_scale, p.green_only = int(cfg.intensity_scale), int(cfg.green_only)
    p.projection = _lib.HG_PROJ[cfg.projection]
    p.pre_relu = int(bool(pre_relu))
    return p, keep


def _ws_bytes(p):
    f, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    check(lib.hg_rgbuv_hist_workspace_bytes(ctypes.byref(p), ctypes.byref(f), ctypes.byref(b)),
          'hg_rgbuv_hist_workspace_bytes')
    return f.value, b.value


def _stream(device):
    return raw_stream(device)


def _require_gpu(x, what):
    if not x.is_cuda:
        raise RuntimeError(f'{what}: input is on {x.device}; the MI355X-native path has no CPU implementation')


class RGBuvHistFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, pre_relu=False):
        _require_gpu(x, 'RGBuvHistFunction')
        x = x.detach()
        if x.dtype != torch.float32:
            x = x.float()
        p, keep = _make_params(x, cfg, pre_relu)
        ctx.pre_relu = pre_relu
        fwd_b, _ = _ws_bytes(p)
        with on_device(x.device):
            # per-pixel projection cache for the backward (32 B per histogram pixel): only when a gradient will be asked
            # for and only when the dense MFMA kernels will run (the scatter paths -- thresholding, narrow RBF --
            # re-classify pixels cheaply and ignore it)
            cache = None
            if ctx.needs_input_grad[0] and PROJ_CACHE and lib.hg_rgbuv_hist_uses_proj_cache(ctypes.byref(p)) == 1:
                cache = torch.empty((p.B, p.Hs * p.Ws, 8), dtype=torch.float32, device=x.device)
                p.proj_cache = cache.data_ptr()
            ctx.cache = cache
            P = 1 if (cfg.green_only or cfg.projection != 'rgbuv') else 3
            out = torch.empty((p.B, P, cfg.h, cfg.h), dtype=torch.float32, device=x.device)
            sums = torch.empty((p.B,), dtype=torch.float32, device=x.device)
            ws = torch.empty((max(fwd_b, 4),), dtype=torch.uint8, device=x.device)
            check(lib.hg_rgbuv_hist_fwd(ctypes.byref(p), x.data_ptr(), out.data_ptr(), sums.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream(x.device)), 'hg_rgbuv_hist_fwd')
        ctx.cfg = cfg
        ctx.save_for_backward(x, out, sums)
        ctx._keep = keep
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, out, sums = ctx.saved_tensors
        cfg = ctx.cfg
        p, keep = _make_params(x, cfg, ctx.pre_relu)
        if ctx.cache is not None:
            p.proj_cache = ctx.cache.data_ptr()
        _, bwd_b = _ws_bytes(p)
        g = grad_out.detach()
        if g.dtype != torch.float32:
            g = g.float()
        g = g.contiguous()
        with on_device(x.device):
            gx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            ws = torch.empty((max(bwd_b, 4),), dtype=torch.uint8, device=x.device)
            check(lib.hg_rgbuv_hist_bwd(ctypes.byref(p), x.data_ptr(), g.data_ptr(), out.data_ptr(),
                                        sums.data_ptr(), gx.data_ptr(), ws.data_ptr(), ws.numel(),
                                        _stream(x.device)), 'hg_rgbuv_hist_bwd')
        return gx, None, None


def run_block(x, cfg, device, what, pre_relu=False):
    """forward() of the drop-in histogram modules: resolves the module's `device` argument the way the reference does
    ('cuda', 'cpu', an ordinal, a torch.device).

    device='cpu' is what the reference's Dataset uses inside forked DataLoader workers (histoGAN/histoGAN.py:263-266,
    296-302): that call runs `hist_cpu.hist_cpu` -- PyTorch CPU ops only, no HIP call, no GPU memory (SURVEY.md
    section 8b: the replacement must not touch the GPU when device='cpu').  It is a separate implementation for that
    contract, not a fallback: a GPU module never takes it, and the HIP functions keep refusing CPU tensors."""
    dev = torch.device('cuda', device) if isinstance(device, int) else torch.device(device)
    if dev.type != 'cuda':
        from .hist_cpu import hist_cpu
        return hist_cpu(x if not x.is_cuda else x.cpu(), cfg, pre_relu)
    if not x.is_cuda:
        x = x.to(dev)
    return rgbuv_hist(x, cfg, pre_relu)


def rgbuv_hist(x, cfg, pre_relu=False):
    """Differentiable RGB-uv histogram of x (B,C>=3,H,W) -> (B, 3|1, h, h), on x's GPU.  pre_relu: the result and
    gradient of `rgbuv_hist(F.relu(x))` (the train step's call, histoGAN/histoGAN.py:955) without the relu launch."""
    return RGBuvHistFunction.apply(x, cfg, pre_relu)


class HellingerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, gen, alpha):
        _require_gpu(gen, 'HellingerFunction')
        t = target.detach().float().contiguous()
        g = gen.detach().float().contiguous()
        if t.shape != g.shape:
            raise ValueError(f'shape mismatch {tuple(t.shape)} vs {tuple(g.shape)}')
        n = g.numel()
        with on_device(g.device):
            loss = torch.empty((), dtype=torch.float32, device=g.device)
            grad = torch.empty_like(g)
            wsb = lib.hg_hellinger_workspace_bytes(n)
            ws = torch.empty((wsb,), dtype=torch.uint8, device=g.device)
            check(lib.hg_hellinger_fwd_bwd(t.data_ptr(), g.data_ptr(), n, g.shape[0], float(alpha),
                                           loss.data_ptr(), grad.data_ptr(), ws.data_ptr(), wsb,
                                           _stream(g.device)), 'hg_hellinger_fwd_bwd')
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        return None, grad * gl, None


class _GlobalHellinger(torch.autograd.Function):
    """The Hellinger loss of the GLOBAL batch under data parallelism: the reference takes ONE square root over the whole
    batch (histoGAN/histoGAN.py:957-960), so the per-rank value sqrt(S_r)/B_r is not a shard of it.  One fp32 all-reduce
    of S_r = sum (sqrt t - sqrt g)^2 gives every rank D = sqrt(sum_r S_r):
        loss = alpha/sqrt2 * D / B_global                                  (the same number on every rank)
        dloss/dg (this rank's samples) = local gradient * (B_r D_r) / (B_global D)
    and because the gradient all-reduce AVERAGES over ranks, the local gradient is scaled by world * that = D_r / D."""

    @staticmethod
    def forward(ctx, local_loss, alpha, batch_local):
        import torch.distributed as dist
        world = dist.get_world_size()
        d_local = local_loss.detach() * (2.0 ** 0.5) * batch_local / alpha          # sqrt(S_r)
        s = (d_local * d_local).reshape(1).clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        d_glob = torch.sqrt(s[0])
        ctx.scale = d_local / d_glob
        return alpha / (2.0 ** 0.5) * d_glob / (batch_local * world)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None, None


HELLINGER_LOCAL = os.environ.get('HG_HELLINGER_LOCAL', '0') != '0'


def hellinger_loss(target_hist, gen_hist, alpha=1.0, global_batch=None):
    """alpha/sqrt(2) * sqrt(sum((sqrt(t)-sqrt(g))^2)) / B  (histoGAN/histoGAN.py:957-960).
    Gradient flows to gen_hist only (the reference's d/d target is computed and never used).
    global_batch: under data parallelism, evaluate the formula on the global batch (one scalar all-reduce; default when
    a process group with more than one rank is initialised, HG_HELLINGER_LOCAL=1 for the per-shard formula)."""
    loss = HellingerFunction.apply(target_hist, gen_hist, alpha)
    if global_batch is None:
        from . import ddp
        global_batch = ddp.is_dist() and not HELLINGER_LOCAL
    if global_batch:
        loss = _GlobalHellinger.apply(loss, float(alpha), int(gen_hist.shape[0]))
    return loss
