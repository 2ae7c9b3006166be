#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8 rows f-2 and f-4, from the UNMODIFIED reference classes.

    python tests/golden/make_golden_dataset.py      # writes tests/golden/dataset.npz + tests/golden/dataset_images/*.png

f-2  The reference's `Dataset` (histoGAN/histoGAN.py:253-307) on the shipped `target_images/*.jpg`: for every index
     `__getitem__` in training mode ({'images', 'histograms' = hist_interpolation of the CPU RGB-uv histograms of two
     random images}) and in test mode (the image's own histogram).  The reference class is imported through the
     sys.modules shims of make_golden_nets.py; ONLY `torchvision.transforms` is restated (PIL calls with torchvision's
     size arithmetic, below) -- the histogram block, hist_interpolation and the Dataset logic are the reference's own.
     The random draws of an item (numpy: the two image indices; torch: the interpolation ratio) are re-played from the
     seed and stored, so a test can feed them to histogan_amd.data.FolderData.
     The decoded pixels of the JPEGs travel as lossless PNGs (the reference tree does not exist on the GPU box).
f-4  `Trainer.generate_truncated` (:1064-1091) called unbound on a stand-in `self` (batch_size, av) with small reference
     networks and fixed latents / noise / av (`.cuda()` patched to a no-op): output images per truncation psi.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = '/root/reference'
IMG_DIR = os.path.join(HERE, 'dataset_images')


# ---- torchvision.transforms restated on PIL (the only restated piece) ------------------------------------------------
def _tv_resize(img, size, interpolation=None):
    """torchvision.transforms.functional.resize(PIL image, int): short side -> size, long side int(size * long / short)
    (truncation), unchanged when the short side already matches; PIL BILINEAR."""
    from PIL import Image
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return img
    new_short, new_long = size, int(size * long / short)
    ow, oh = (new_short, new_long) if w <= h else (new_long, new_short)
    return img.resize((ow, oh), Image.BILINEAR)


def _tv_center_crop(img, size):
    """torchvision CenterCrop: top/left = int(round((dim - size) / 2.0)) (Python round: half to even)."""
    w, h = img.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def _tv_to_tensor(img):
    arr = np.asarray(img, dtype=np.uint8)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous().float().div(255)


def make_transforms_module():
    T = types.ModuleType('torchvision.transforms')

    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Lambda:
        def __init__(self, fn): self.fn = fn
        def __call__(self, x): return self.fn(x)

    class Resize:
        def __init__(self, size): self.size = size
        def __call__(self, x): return _tv_resize(x, self.size)

    class CenterCrop:
        def __init__(self, size): self.size = size
        def __call__(self, x): return _tv_center_crop(x, self.size)

    class ToTensor:
        def __call__(self, x): return _tv_to_tensor(x)

    class RandomResizedCrop:                       # constructed by the reference's transform, only called when aug_prob > 0
        def __init__(self, size, scale=None, ratio=None): self.size, self.scale, self.ratio = size, scale, ratio
        def __call__(self, x): raise NotImplementedError('aug_prob = 0 in the golden run')

    T.Compose, T.Lambda, T.Resize, T.CenterCrop, T.ToTensor, T.RandomResizedCrop = \
        Compose, Lambda, Resize, CenterCrop, ToTensor, RandomResizedCrop
    F = types.ModuleType('torchvision.transforms.functional')
    F.resize = _tv_resize
    T.functional = F
    return T, F


def import_reference():
    from make_golden_nets import import_reference as imp
    T, F = make_transforms_module()
    # make_golden_nets installs empty stubs; put the restated transforms in their place before the reference imports them
    R = None
    import make_golden_nets as M
    orig = M.types.ModuleType

    def patched(name):
        if name == 'torchvision.transforms':
            return T
        return orig(name)
    M.types.ModuleType = patched
    try:
        R = imp()
    finally:
        M.types.ModuleType = orig
    sys.modules['torchvision.transforms.functional'] = F
    sys.modules['torchvision'].transforms = T
    return R


def main():
    from PIL import Image
    R = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}

    # ---- f-2: the reference Dataset on target_images (decoded pixels re-saved as PNG: identical arrays)
    os.makedirs(IMG_DIR, exist_ok=True)
    names = ['2', '3', '5', '6', '1']                     # 256^2 x3, 799x533 (non-square), 1024^2
    for n in names:
        Image.open(os.path.join(REF, 'target_images', f'{n}.jpg')).convert('RGB').save(os.path.join(IMG_DIR, f't{n}.png'),
                                                                                        optimize=True)
    S, HB, INSZ = 256, 64, 150
    ds = R.Dataset(IMG_DIR, image_size=S, hist_insz=INSZ, hist_bin=HB, hist_method='inverse-quadratic',
                   hist_resizing='sampling')
    ds.paths = sorted(ds.paths)                           # glob order is file-system order: pin it
    dt = R.Dataset(IMG_DIR, image_size=S, hist_insz=INSZ, hist_bin=HB, hist_method='inverse-quadratic',
                   hist_resizing='sampling', test=True)
    dt.paths = sorted(dt.paths)
    out['ds_paths'] = np.array([os.path.basename(str(p)) for p in ds.paths])
    out['ds_meta'] = np.array([S, HB, INSZ])
    for i in range(len(ds.paths)):
        seed = 1000 + i
        np.random.seed(seed); torch.manual_seed(seed)
        item = ds[i]
        np.random.seed(seed); torch.manual_seed(seed)
        inds = np.random.randint(0, high=len(ds.paths), size=2)     # the draws __getitem__ made (:297, :181)
        ratio = torch.rand(1)
        img = item['images']
        u8 = (img * 255).round().to(torch.uint8)
        assert torch.equal(u8.float().div(255), img)                # ToTensor output is k/255 exactly
        out[f'ds_item{i}_images_u8'] = u8.numpy()
        out[f'ds_item{i}_histograms'] = item['histograms'].numpy()
        out[f'ds_item{i}_inds'] = inds
        out[f'ds_item{i}_ratio'] = ratio.numpy()
        out[f'ds_own{i}'] = dt[i]['histograms'].numpy()
        print(i, out['ds_paths'][i], tuple(img.shape), inds, float(ratio), float(item['histograms'].sum()))
    # a second histogram configuration of the same flow: bilinear resize to insz (hist_resizing='interpolation')
    db = R.Dataset(IMG_DIR, image_size=S, hist_insz=INSZ, hist_bin=HB, hist_method='inverse-quadratic',
                   hist_resizing='interpolation', test=True)
    db.paths = sorted(db.paths)
    for i in range(len(db.paths)):
        out[f'ds_own_interp{i}'] = db[i]['histograms'].numpy()

    # ---- f-4: generate_truncated of the reference Trainer, unbound, on small reference networks
    torch.manual_seed(7)
    S_, CAP, LAT, HBs, NT = 32, 4, 32, 16, 2
    G = R.Generator(S_, LAT, network_capacity=CAP)
    SV = R.StyleVectorizer(LAT, 3)
    HV = R.HistVectorizer(HBs, LAT, 3)
    for blk in G.blocks:
        for lin in (blk.to_noise1, blk.to_noise2):
            torch.nn.init.normal_(lin.weight, std=0.5)
            torch.nn.init.normal_(lin.bias, std=0.1)
    for prefix, m in (('gt_G', G), ('gt_S', SV), ('gt_H', HV)):
        for k, v in m.state_dict().items():
            out[f'{prefix}/{k}'] = v.detach().numpy()
    n_img = NT * NT
    z = torch.randn(n_img, LAT)
    noi = torch.rand(n_img, S_, S_, 1)
    hist = torch.rand(1, 3, HBs, HBs); hist = hist / hist.sum()
    hb = hist
    for _ in range(int(np.log2(NT))):                       # the CLI's duplication (histoGAN.py:122-125)
        hb = torch.cat((hb, hb), dim=0)
    av = SV(torch.randn(64, LAT)).detach().numpy().mean(axis=0, keepdims=True)
    out.update(gt_meta=np.array([S_, CAP, LAT, HBs, NT]), gt_z=z.numpy(), gt_noise=noi.numpy(), gt_hist=hb.numpy(), gt_av=av)
    for psi in (0.75, 0.4):
        fake_self = types.SimpleNamespace(av=av.copy(), batch_size=3)      # batch_size 3: evaluate_in_chunks splits 4 -> 3 + 1
        with torch.no_grad():
            imgs = R.Trainer.generate_truncated(fake_self, SV, HV, G, hb, [(z, G.num_layers - 2)], noi, trunc_psi=psi)
        out[f'gt_images_psi{psi}'] = imgs.numpy()
        print('generate_truncated psi', psi, tuple(imgs.shape), float(imgs.mean()))

    np.savez_compressed(os.path.join(HERE, 'dataset.npz'), **out)
    print('wrote', os.path.join(HERE, 'dataset.npz'), os.path.getsize(os.path.join(HERE, 'dataset.npz')) // 1024, 'KiB')
    print({f: os.path.getsize(os.path.join(IMG_DIR, f)) // 1024 for f in sorted(os.listdir(IMG_DIR))})


if __name__ == '__main__':
    main()
