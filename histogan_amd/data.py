"""Minimal image-folder source and image-grid writer (PIL only; torchvision is absent in this image).

The reference's Dataset (histoGAN/histoGAN.py:253-307) decodes with PIL/torchvision in DataLoader
workers and computes two CPU target histograms per item; that pipeline is outside the hot-path scope
(SURVEY.md section 8f row f-2).  This source keeps the same item contract -- {'images': (B,3,S,S) in
[0,1], 'histograms': interpolation of the histograms of two other random images} -- but computes the
target histograms as ONE batched GPU call per batch.
"""
from pathlib import Path

import numpy as np
import torch

EXTS = ['jpg', 'png']


def _load_rgb(path, size=None):
    from PIL import Image
    img = Image.open(path).convert('RGB')
    if size is not None:
        w, h = img.size
        s = size / min(w, h)                                   # transforms.Resize(size): short side -> size
        img = img.resize((max(size, round(w * s)), max(size, round(h * s))), Image.BILINEAR)
        w, h = img.size
        l, t = (w - size) // 2, (h - size) // 2                # transforms.CenterCrop(size)
        img = img.crop((l, t, l + size, t + size))
    return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()  # ToTensor


class FolderData:
    def __init__(self, folder, hist_block, batch_size, image_size, device, transparent=False, seed=0, test=False,
                 hist_sampling=True):
        if transparent:
            raise NotImplementedError('transparent (RGBA) images are not supported')
        self.paths = [p for ext in EXTS for p in Path(f'{folder}').glob(f'**/*.{ext}')]
        if not self.paths:
            raise FileNotFoundError(f'no {EXTS} images under {folder}')
        self.hist_block, self.B, self.S, self.device, self.test = hist_block, batch_size, image_size, device, test
        self.rs = np.random.RandomState(seed)
        # True: target = interpolation of the histograms of two OTHER random images; False: the image's own
        # histogram (ReHistoGAN/rehistoGAN.py:375-446, `hist_sampling`)
        self.hist_sampling = hist_sampling

    def _hist_of(self, idx):
        # full-resolution images differ in size: one GPU call per image, still no CPU histogram
        hs = []
        for i in idx:
            x = _load_rgb(self.paths[i]).unsqueeze(0).to(self.device)
            with torch.no_grad():
                hs.append(self.hist_block(x))
        return torch.cat(hs, 0)

    def __iter__(self):
        return self

    def __next__(self):
        n = len(self.paths)
        if self.test:
            return {'histograms': self._hist_of(self.rs.randint(0, n, self.B))}
        idx = self.rs.randint(0, n, self.B)
        images = torch.stack([_load_rgb(self.paths[i], self.S) for i in idx]).to(self.device)
        if not self.hist_sampling:
            return {'images': images, 'histograms': self._hist_of(idx)}
        h1 = self._hist_of(self.rs.randint(0, n, self.B))
        h2 = self._hist_of(self.rs.randint(0, n, self.B))
        ratio = torch.rand(self.B, 1, 1, 1, device=self.device)   # hist_interpolation (:180-182), per item
        return {'images': images, 'histograms': h1 * ratio + h2 * (1 - ratio)}


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
