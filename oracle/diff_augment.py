"""CPU restatement of DiffAugment with EXPLICIT per-sample parameters.  TEST INFRASTRUCTURE ONLY.

The reference (utils/diff_augment.py:9-107, histoGAN/histoGAN.py:312-315) draws its random parameters inside each
function; here the parameters are arguments, so the same table can drive the reference's functions (RNG patched,
tests/golden/make_golden_augment.py), this restatement and the HIP kernels.  Pinned by tests/test_oracle_augment_golden.py.

Row layout (include/hg_augment.h): [flip, roll_h, roll_w, shift_h, shift_w, r0, r1, c0, c1], applied in that order.
"""
import torch


def spatial(x, params):
    """x (B,C,H,W) float; params (B,9) ints."""
    B, C, H, W = x.shape
    out = []
    for b in range(B):
        flip, rh, rw, sh, sw, r0, r1, c0, c1 = [int(v) for v in params[b]]
        img = x[b]
        if flip:
            img = torch.flip(img, dims=(2,))                     # random_hflip: dims=(3,) of the batch
        if rw:
            img = torch.roll(img, rw, 2)                         # rand_offset :63-64
        if rh:
            img = torch.roll(img, rh, 1)                         # :66-67
        if sh or sw:                                             # rand_translation :33-50: zero-filled shift
            pad = torch.zeros(C, H + 2 * abs(sh), W + 2 * abs(sw), dtype=x.dtype)
            pad[:, abs(sh):abs(sh) + H, abs(sw):abs(sw) + W] = img
            img = pad[:, abs(sh) + sh:abs(sh) + sh + H, abs(sw) + sw:abs(sw) + sw + W]
        if r0 <= r1 and c0 <= c1:                                # rand_cutout :78-97
            mask = torch.ones(H, W, dtype=x.dtype)
            mask[r0:r1 + 1, c0:c1 + 1] = 0
            img = img * mask
        out.append(img)
    return torch.stack(out)


def color(x, col):
    """col (B,3): brightness offset, saturation factor, contrast factor; reference order :16-31."""
    col = torch.as_tensor(col, dtype=x.dtype)
    x = x + col[:, 0].view(-1, 1, 1, 1)
    m = x.mean(dim=1, keepdim=True)
    x = (x - m) * col[:, 1].view(-1, 1, 1, 1) + m
    m = x.mean(dim=[1, 2, 3], keepdim=True)
    return (x - m) * col[:, 2].view(-1, 1, 1, 1) + m


def cutout_box(off_h, off_w, H, W, ratio=0.5):
    """The index ranges rand_cutout zeroes for offsets (off_h, off_w): clamp of a contiguous range (:80-93)."""
    ch, cw = int(H * ratio + 0.5), int(W * ratio + 0.5)
    cl = lambda v, n: max(0, min(n - 1, v))
    return (cl(off_h - ch // 2, H), cl(off_h - ch // 2 + ch - 1, H), cl(off_w - cw // 2, W), cl(off_w - cw // 2 + cw - 1, W))
