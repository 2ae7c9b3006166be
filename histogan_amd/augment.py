"""DiffAugment on the MI355X kernels of include/hg_augment.h.

Mirror of utils/diff_augment.py (reference :9-107) and of AugWrapper / random_hflip (histoGAN/histoGAN.py:312-331):
same augmentation names, same parameter distributions, same order of application.  The random parameters of a
whole chain are drawn on the host (one small table per batch, one H2D copy); consecutive spatial augmentations
(flip, offset*, translation, cutout -- in that order) share ONE launch, the colour ones another.  Every augmentation
is linear in the image, its backward is the adjoint kernel wrapped as a Function again, so the chain is
differentiable to any order (the gradient penalty differentiates twice through the augmented real images).
"""
from random import random

import torch
from torch import nn

from ._lib import check, lib, on_device, raw_stream
from .ops import _f32c, _need_gpu, _st

NP = 9                                    # HG_AUG_NP
_ID = (0, 0, 0, 0, 0, 1, 0, 1, 0)         # identity row: no flip/roll/shift, empty cutout (r0 > r1)


class _Spatial(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, adjoint):
        _need_gpu(x, 'DiffAugment')
        x = _f32c(x.detach())
        B, C, H, W = x.shape
        with on_device(x.device):
            out = torch.empty_like(x)
            check(lib.hg_augment_spatial(x.data_ptr(), params.data_ptr(), out.data_ptr(), B, C, H, W, int(adjoint),
                                         _st(x)), 'hg_augment_spatial')
        ctx.params, ctx.adjoint = params, adjoint
        return out

    @staticmethod
    def backward(ctx, g):
        return _Spatial.apply(g, ctx.params, not ctx.adjoint), None, None


def _sample_mean(x):
    B = x.shape[0]
    n = lib.hg_augment_workspace_bytes(B)
    ws = torch.empty(max(n, 4), dtype=torch.uint8, device=x.device)
    mean = torch.empty(B, dtype=torch.float32, device=x.device)
    check(lib.hg_sample_mean(x.data_ptr(), mean.data_ptr(), B, x[0].numel(), ws.data_ptr(), n, _st(x)), 'hg_sample_mean')
    return mean


class _Color(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, color, adjoint):
        _need_gpu(x, 'DiffAugment')
        x = _f32c(x.detach())
        B, C, H, W = x.shape
        with on_device(x.device):
            out = torch.empty_like(x)
            check(lib.hg_augment_color(x.data_ptr(), _sample_mean(x).data_ptr(), color.data_ptr(), out.data_ptr(), B, C,
                                       H * W, int(adjoint), _st(x)), 'hg_augment_color')
        ctx.color, ctx.adjoint = color, adjoint
        return out

    @staticmethod
    def backward(ctx, g):
        # derivatives see the LINEAR part only: the brightness offset is dropped
        lin = ctx.color.clone()
        lin[:, 0] = 0
        return _Color.apply(g, lin, not ctx.adjoint), None, None


def augment_spatial(x, params):
    """x (B,C,H,W); params: (B, 9) int32 rows [flip, roll_h, roll_w, shift_h, shift_w, r0, r1, c0, c1] (include/hg_augment.h)."""
    p = torch.as_tensor(params, dtype=torch.int32).reshape(x.shape[0], NP).clone()
    p[:, 1] %= x.shape[2]
    p[:, 2] %= x.shape[3]
    return _Spatial.apply(x, p.to(x.device).contiguous(), False)


def augment_color(x, color):
    """x (B,C,H,W); color: (B, 3) rows [brightness offset, saturation factor, contrast factor]."""
    c = torch.as_tensor(color, dtype=torch.float32).reshape(x.shape[0], 3)
    return _Color.apply(x, c.to(x.device).contiguous(), False)


# ---- parameter draws: the distributions of utils/diff_augment.py, on the host --------------------------------
def _gen(generator):
    return dict(generator=generator) if generator is not None else {}


def draw_translation(B, H, W, ratio=0.125, generator=None):
    sh, sw = int(H * ratio + 0.5), int(W * ratio + 0.5)
    return (torch.randint(-sh, sh + 1, (B,), **_gen(generator)), torch.randint(-sw, sw + 1, (B,), **_gen(generator)))


def draw_cutout(B, H, W, ratio=0.5, generator=None):
    """-> r0, r1, c0, c1 (inclusive), the clamped index ranges the reference zeroes (diff_augment.py:78-97)."""
    ch, cw = int(H * ratio + 0.5), int(W * ratio + 0.5)
    off_h = torch.randint(0, H + (1 - ch % 2), (B,), **_gen(generator))
    off_w = torch.randint(0, W + (1 - cw % 2), (B,), **_gen(generator))
    r0 = (off_h - ch // 2).clamp(0, H - 1)
    r1 = (off_h - ch // 2 + ch - 1).clamp(0, H - 1)
    c0 = (off_w - cw // 2).clamp(0, W - 1)
    c1 = (off_w - cw // 2 + cw - 1).clamp(0, W - 1)
    return r0, r1, c0, c1


def draw_offset(B, H, W, ratio=1, ratio_h=1, ratio_v=1, generator=None):
    """-> roll of dim W, roll of dim H.  As in the reference (diff_augment.py:52-70) the roll of W is bounded by
    int(H * ratio * ratio_h) and the roll of H by int(W * ratio * ratio_v) (its w/h names are swapped)."""
    max_h, max_v = int(H * ratio * ratio_h), int(W * ratio * ratio_v)
    vh = torch.randint(0, max_h + 1, (B,), **_gen(generator)) * 2 - max_h
    vv = torch.randint(0, max_v + 1, (B,), **_gen(generator)) * 2 - max_v
    return vh, vv


SPATIAL = ('offset', 'offset_h', 'offset_v', 'translation', 'cutout')
_RANK = {'flip': 0, 'offset': 1, 'offset_h': 1, 'offset_v': 1, 'translation': 2, 'cutout': 3}


def _spatial_run(x, names, flip=None, generator=None, ratios=None):
    """One launch for `names` (non-decreasing _RANK order, each rank at most once), optionally preceded by a flip.
    ratios: optional {name: dict of the reference function's ratio arguments} (defaults: utils/diff_augment.py)."""
    B, _, H, W = x.shape
    ratios = ratios or {}
    p = torch.tensor(_ID, dtype=torch.int32).repeat(B, 1)
    if flip is not None:
        p[:, 0] = flip
    for n in names:
        kw = ratios.get(n, {})
        if n in ('offset', 'offset_h', 'offset_v'):
            vh, vv = draw_offset(B, H, W, kw.get('ratio', 1), kw.get('ratio_h', 0 if n == 'offset_v' else 1),
                                 kw.get('ratio_v', 0 if n == 'offset_h' else 1), generator)
            p[:, 2], p[:, 1] = vh.int(), vv.int()
        elif n == 'translation':
            sh, sw = draw_translation(B, H, W, kw.get('ratio', 0.125), generator=generator)
            p[:, 3], p[:, 4] = sh.int(), sw.int()
        elif n == 'cutout':
            for k, v in zip((5, 6, 7, 8), draw_cutout(B, H, W, kw.get('ratio', 0.5), generator=generator)):
                p[:, k] = v.int()
    return augment_spatial(x, p)


def DiffAugment(x, types=[], flip=None, generator=None):
    """utils/diff_augment.py:9-13: apply the augmentations named in `types`, in that order.
    flip: optional per-sample 0/1 tensor (or bool) applied first (AugWrapper's random_hflip) in the same launch."""
    if not x.is_cuda:
        raise RuntimeError('DiffAugment: the MI355X-native path has no CPU implementation')
    B = x.shape[0]
    run, last = [], -1
    pending_flip = None if flip is None else torch.as_tensor(flip, dtype=torch.int32).expand(B)

    def flush():
        nonlocal run, last, pending_flip, x
        if run or pending_flip is not None:
            x = _spatial_run(x, run, pending_flip, generator)
        run, last, pending_flip = [], -1, None

    for t in types:
        if t == 'color':
            flush()
            g = _gen(generator)
            color = torch.stack([torch.rand(B, **g) - 0.5, torch.rand(B, **g) * 2, torch.rand(B, **g) + 0.5], dim=1)
            x = augment_color(x, color)
        elif t in SPATIAL:
            if _RANK[t] <= last:
                flush()
            run.append(t)
            last = _RANK[t]
        else:
            raise KeyError(t)
    flush()
    return x.contiguous()


def random_hflip(tensor, prob):
    """histoGAN/histoGAN.py:312-315 (the whole batch is flipped or not)."""
    if prob > random():
        return tensor
    return DiffAugment(tensor, [], flip=1)


class AugWrapper(nn.Module):
    """histoGAN/histoGAN.py:318-331: with probability `prob`, flip (p=0.5) + DiffAugment the batch entering D."""

    def __init__(self, D):
        super().__init__()
        self.D = D

    def augment(self, images, prob=0.0, types=[], detach=False):
        if random() < prob:
            flip = 0 if 0.5 > random() else 1
            images = DiffAugment(images, types=types, flip=flip)
        return images.detach() if detach else images

    def forward(self, images, prob=0.0, types=[], detach=False):
        return self.D(self.augment(images, prob, types, detach))
