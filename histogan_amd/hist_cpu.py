"""`device='cpu'` of the drop-in histogram modules: a HIP-free, batch-vectorised implementation on PyTorch's CPU ops.

The reference constructs `RGBuvHistBlock(..., device='cpu')` inside its Dataset and calls it from forked DataLoader
workers (histoGAN/histoGAN.py:263-266, 296-302; ctor histogram_classes/RGBuvHistBlock.py:29-31), so this path must not
touch the GPU or any HIP runtime state: it imports nothing but torch and numpy (not `_lib`, not the `.so`).  It is NOT
the measured path and NOT a fallback of it: GPU tensors never come here, CPU tensors never reach the HIP kernels
(`hist.run_block` routes on the module's `device` argument only).

Same arithmetic types as the reference chain (RGBuvHistBlock.py:75-228; rgChromaHistBlock.py:73-145; LabHistBlock.py:73-144):
fp32 clamp / resize / projection, fp64 bin distances and kernel values, fp32 accumulation -- but one batched pass instead
of the per-image Python loop: the projection is evaluated once per pixel (3 logarithms instead of 12), the two kernel
matrices of a plane are built for the whole chunk of images at once and contracted with one `bmm`; memory is bounded by
processing `chunk` images at a time.  Differentiable through autograd (the reference's own mechanism on the CPU)."""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-6
_CHUNK_BYTES = 256 << 20          # fp64 kernel matrices of one chunk: 2 * chunk * N * h * 8 bytes stay below this


def _resize(x, cfg):
    """clamp + resize stage (RGBuvHistBlock.py:76-99): bilinear to insz x insz, or h x h strided samples."""
    x = torch.clamp(x, 0, 1)
    if x.shape[2] > cfg.insz or x.shape[3] > cfg.insz:
        if cfg.resizing == 'interpolation':
            x = F.interpolate(x, size=(cfg.insz, cfg.insz), mode='bilinear', align_corners=False)
        elif cfg.resizing == 'sampling':
            r = torch.from_numpy(np.linspace(0, x.shape[2], cfg.h, endpoint=False).astype(np.int64))
            c = torch.from_numpy(np.linspace(0, x.shape[3], cfg.h, endpoint=False).astype(np.int64))
            x = x.index_select(2, r).index_select(3, c)
        else:
            raise Exception(f'Wrong resizing method. It should be: interpolation or sampling. '
                            f'But the given value is {cfg.resizing}.')
    return x[:, :3]


def _planes(I, cfg):
    """Pixel weight and the (u, v) coordinate pairs of every histogram plane.  I: (b, 3, N) fp32."""
    if cfg.projection == 'rgbuv':
        w = torch.sqrt((I * I).sum(dim=1) + EPS) if cfg.intensity_scale else None
        L = torch.log(I + EPS)
        r, g, b = L[:, 0], L[:, 1], L[:, 2]
        green = (g - r, g - b)
        return w, [green] if cfg.green_only else [(r - g, r - b), green, (b - r, b - g)]
    if cfg.projection == 'rgchroma':
        w = torch.sqrt((I * I).sum(dim=1) + EPS) if cfg.intensity_scale else None
        s = I.sum(dim=1) + EPS
        return w, [(I[:, 0] / s, I[:, 1] / s)]
    if cfg.projection == 'direct':                 # Lab: channel 0 weighs, channels 1 / 2 are the coordinates
        return (I[:, 0] if cfg.intensity_scale else None), [(I[:, 1], I[:, 2])]
    raise ValueError(f'unknown projection {cfg.projection!r}')


def _kernel(coord, bins, cfg):
    """(b, N) fp32 coordinates -> (b, N, h) fp32 soft-bin weights, evaluated in fp64 like the reference (:116-146)."""
    d = (coord.unsqueeze(-1) - bins).abs()         # fp32 - fp64 -> fp64
    if cfg.method == 'thresholding':
        eps = (abs(cfg.lo) + abs(cfg.hi)) / cfg.h
        k = d <= eps / 2
    elif cfg.method == 'RBF':
        k = torch.exp(-(d * d) / cfg.sigma ** 2)
    elif cfg.method == 'inverse-quadratic':
        k = 1 / (1 + (d * d) / cfg.sigma ** 2)
    else:
        raise Exception(f'Wrong kernel method. It should be either thresholding, RBF,'
                        f' inverse-quadratic. But the given value is {cfg.method}.')
    return k.to(torch.float32)


def hist_cpu(x, cfg, pre_relu=False):
    """x: CPU float (B, C>=3, H, W) -> CPU float32 (B, 3|1, h, h), L1-normalised per image (:224-228)."""
    if x.is_cuda:
        raise RuntimeError('hist_cpu: GPU tensor on the CPU path')
    if x.dim() != 4 or x.shape[1] < 3:
        raise ValueError(f'expected (B, C>=3, H, W) input, got {tuple(x.shape)}')
    if x.dtype != torch.float32:
        x = x.float()
    if pre_relu:
        x = F.relu(x)
    xs = _resize(x, cfg)
    B, N = xs.shape[0], xs.shape[2] * xs.shape[3]
    I = xs.reshape(B, 3, N)
    bins = torch.from_numpy(np.linspace(cfg.lo, cfg.hi, num=cfg.h))
    chunk = max(1, _CHUNK_BYTES // max(1, 16 * N * cfg.h))
    parts = []
    for s in range(0, B, chunk):
        w, planes = _planes(I[s:s + chunk], cfg)
        hs = []
        for u, v in planes:
            ku, kv = _kernel(u, bins, cfg), _kernel(v, bins, cfg)
            if w is not None:
                ku = ku * w.unsqueeze(-1)
            hs.append(torch.bmm(ku.transpose(1, 2), kv))
        parts.append(torch.stack(hs, dim=1))
    hists = torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]
    return hists / (hists.sum(dim=(1, 2, 3)).view(-1, 1, 1, 1) + EPS)
