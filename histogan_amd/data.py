"""Image-folder source and image-grid writer (PIL only; torchvision is absent in this image).

The reference's Dataset (histoGAN/histoGAN.py:253-307, ReHistoGAN/rehistoGAN.py:335-446) decodes with PIL/torchvision in
DataLoader workers and computes TWO CPU target histograms of full-resolution images per item (~1 s each on the
reference's fp64 CPU path): at the 470 images/s of the GPU step that pipeline would starve the GPU by three orders of
magnitude.  This source keeps the item contract -- {'images': (B,3,S,S) in [0,1],
'histograms': interpolation of the histograms of two other random images, or the image's own} -- and restructures it:

* decoding / resizing runs in a thread pool (PIL releases the GIL) and batches are prefetched `prefetch` deep;
* target histograms are computed on the GPU (one launch per full-resolution image, sizes differ) and CACHED per
  image on the device: the histogram of image i never changes, so after the first pass over the folder a batch's
  targets are two gathers and one interpolation -- the reference's "precompute histograms.npy" flow
  (create_hist_data.py:33-55) done lazily;
* nothing touches the CPU histogram path.
"""
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

EXTS = ['jpg', 'png']


def _random_resized_crop_box(w, h, rs, scale=(0.5, 1.0), ratio=(0.98, 1.02)):
    """torchvision.transforms.RandomResizedCrop.get_params restated (histoGAN/histoGAN.py:277-279 uses scale (0.5, 1),
    ratio (0.98, 1.02)): up to 10 draws of (area fraction, log-uniform aspect), then the ratio-clamped centre crop."""
    area = w * h
    lr = np.log(ratio)
    for _ in range(10):
        target = area * rs.uniform(scale[0], scale[1])
        ar = np.exp(rs.uniform(lr[0], lr[1]))
        cw, ch = int(round(np.sqrt(target * ar))), int(round(np.sqrt(target / ar)))
        if 0 < cw <= w and 0 < ch <= h:
            return int(rs.randint(0, w - cw + 1)), int(rs.randint(0, h - ch + 1)), cw, ch
    in_ratio = w / h
    if in_ratio < ratio[0]:
        cw, ch = w, int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        cw, ch = int(round(h * ratio[1])), h
    else:
        cw, ch = w, h
    return (w - cw) // 2, (h - ch) // 2, cw, ch


def _load_rgb(path, size=None, flip=False, rgba=False, crop_seed=None):
    """crop_seed: None = Resize + CenterCrop; an int = Resize + RandomResizedCrop(size, scale (0.5,1), ratio (0.98,1.02))
    drawn from that seed (the `dataset_aug_prob` branch of the reference's transform, histoGAN/histoGAN.py:273-283)."""
    from PIL import Image
    img = Image.open(path).convert('RGBA' if rgba else 'RGB')    # convert_rgb_to_transparent / _transparent_to_rgb
    if size is not None:
        w, h = img.size
        # transforms.Resize(size) (after resize_to_minimum_size, which is the same call for small images, :246-249): short
        # side -> size, long side int(size * long / short) -- torchvision TRUNCATES -- and no-op when the short side matches
        short, long = (w, h) if w <= h else (h, w)
        if short != size:
            new_long = int(size * long / short)
            img = img.resize((size, new_long) if w <= h else (new_long, size), Image.BILINEAR)
        w, h = img.size
        if crop_seed is None:
            # transforms.CenterCrop(size): torchvision's offsets are int(round((dim - size) / 2.0)) (round half to even)
            l, t = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
            img = img.crop((l, t, l + size, t + size))
        else:
            l, t, cw, ch = _random_resized_crop_box(w, h, np.random.RandomState(crop_seed))
            img = img.resize((size, size), Image.BILINEAR, box=(l, t, l + cw, t + ch))
    arr = np.asarray(img, dtype=np.float32) / 255.0            # ToTensor
    if flip:
        arr = arr[:, ::-1]                                     # transforms.RandomHorizontalFlip
    return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()


class FolderData:
    def __init__(self, folder, hist_block, batch_size, image_size, device, transparent=False, seed=0, test=False,
                 hist_sampling=True, workers=8, prefetch=3, cache_hists=True, max_cached=200000, hflip=False,
                 aug_prob=0.0, max_cache_bytes=2 << 30):
        self.rgba = bool(transparent)                            # 4-channel items; the histogram uses channels 0..2
        self.paths = sorted(p for ext in EXTS for p in Path(f'{folder}').glob(f'**/*.{ext}'))
        if not self.paths:
            raise FileNotFoundError(f'no {EXTS} images under {folder}')
        self.hist_block, self.B, self.S, self.device, self.test = hist_block, batch_size, image_size, device, test
        self.rs = np.random.RandomState(seed)
        # True: target = interpolation of the histograms of two OTHER random images; False: the image's own
        # histogram (ReHistoGAN/rehistoGAN.py:375-446, `hist_sampling`)
        self.hist_sampling = hist_sampling
        self.hflip = hflip                                      # the reference's training transform flips with p = 0.5
        self.aug_prob = float(aug_prob)                         # `dataset_aug_prob`: P(RandomResizedCrop instead of CenterCrop)
        self.cache = {} if cache_hists else None
        self.max_cached = max_cached
        self.max_cache_bytes, self.cache_bytes = int(max_cache_bytes), 0   # device memory the cached histograms may take
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.queue = deque()
        self.prefetch = max(1, prefetch)
        self.hits = self.misses = 0

    # ---- planning (host RNG, main thread: deterministic for a seed) and decoding (worker threads)
    def _plan(self):
        n = len(self.paths)
        plan = {'img': self.rs.randint(0, n, self.B)}
        own = self.test or not self.hist_sampling              # target = the image's own histogram
        if not own:
            plan['h1'], plan['h2'] = self.rs.randint(0, n, self.B), self.rs.randint(0, n, self.B)
            plan['ratio'] = self.rs.rand(self.B).astype(np.float32)       # hist_interpolation (:180-182), per item
        flips = self.rs.rand(self.B) < 0.5 if (self.hflip and not self.test) else np.zeros(self.B, bool)
        need = set(int(i) for key in (('img',) if own else ('h1', 'h2')) for i in plan[key])
        if self.cache is not None:
            need = {i for i in need if i not in self.cache}
        plan['full'] = {i: self.pool.submit(_load_rgb, self.paths[i], None, False, self.rgba) for i in sorted(need)}   # full resolution
        crops = [int(self.rs.randint(0, 2 ** 31 - 1)) if (self.aug_prob > 0.0 and not self.test and self.rs.rand() < self.aug_prob)
                 else None for _ in range(self.B)]
        plan['small'] = None if self.test else [self.pool.submit(_load_rgb, self.paths[int(i)], self.S, bool(f), self.rgba, c)
                                                for i, f, c in zip(plan['img'], flips, crops)]
        return plan

    def _hist(self, idx, plan):
        """(len(idx), 3, h, h) target histograms on the device: cached, or one GPU launch per missing image."""
        out = []
        for i in (int(v) for v in idx):
            h = self.cache.get(i) if self.cache is not None else None
            if h is None:
                fut = plan['full'].get(i)
                x = (fut.result() if fut is not None else _load_rgb(self.paths[i], None, False, self.rgba)).unsqueeze(0).to(self.device)
                with torch.no_grad():
                    h = self.hist_block(x)[0]
                self.misses += 1
                nbytes = h.numel() * h.element_size()
                if self.cache is not None and len(self.cache) < self.max_cached and \
                        self.cache_bytes + nbytes <= self.max_cache_bytes:
                    self.cache[i] = h
                    self.cache_bytes += nbytes
            else:
                self.hits += 1
            out.append(h)
        return torch.stack(out)

    def __iter__(self):
        return self

    def __next__(self):
        while len(self.queue) < self.prefetch:
            self.queue.append(self._plan())
        plan = self.queue.popleft()
        batch = {}
        if plan['small'] is not None:
            host = torch.stack([f.result() for f in plan['small']])
            batch['images'] = host.pin_memory().to(self.device, non_blocking=True) if self.device.type == 'cuda' \
                else host.to(self.device)
        if 'h1' in plan:
            ratio = torch.from_numpy(plan['ratio']).to(self.device).view(-1, 1, 1, 1)
            batch['histograms'] = self._hist(plan['h1'], plan) * ratio + self._hist(plan['h2'], plan) * (1 - ratio)
        else:
            batch['histograms'] = self._hist(plan['img'], plan)
        return batch


def save_image_grid(images, path, nrow=4, padding=2):
    """torchvision.utils.save_image restated: (N,C,H,W) in [0,1] -> one padded grid image."""
    from PIL import Image
    x = images.detach().float().clamp(0, 1).cpu()
    n, c, h, w = x.shape
    ncol = min(nrow, n)
    nr = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nr * (h + padding) + padding, ncol * (w + padding) + padding)
    for k in range(n):
        r, cc = divmod(k, ncol)
        y0, x0 = padding + r * (h + padding), padding + cc * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = x[k]
    arr = (grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy())
    Image.fromarray(arr[..., :3] if c >= 3 else arr[..., 0]).save(path)
