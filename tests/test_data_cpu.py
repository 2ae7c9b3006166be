"""Image-folder source: item contract, determinism, histogram cache, prefetch.
The histogram block is a stand-in here (no GPU); tests/test_trainer_io_gpu.py runs the real one."""
import os

import numpy as np
import pytest
import torch

from histogan_amd.data import FolderData, _load_rgb, save_image_grid


class FakeHist:
    def __init__(self):
        self.calls = 0

    def __call__(self, x):
        self.calls += 1
        assert x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 3        # ONE full-resolution image per call
        return x.mean(dim=(2, 3)).view(1, 3, 1, 1).expand(1, 3, 4, 4) / 3.0


class FakeHistAny(FakeHist):
    def __call__(self, x):
        return x[:, :3].mean(dim=(2, 3)).view(1, 3, 1, 1).expand(1, 3, 4, 4) / 3.0


@pytest.fixture()
def folder(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    for i in range(7):
        arr = (rs.rand(30 + i, 40 + 2 * i, 3) * 255).astype(np.uint8)
        Image.fromarray(arr).save(os.path.join(tmp_path, f'im{i}.png' if i % 2 else f'im{i}.jpg'))
    Image.fromarray((rs.rand(20, 20) * 255).astype(np.uint8)).save(os.path.join(tmp_path, 'grey.png'))   # greyscale
    return str(tmp_path)


def test_contract_determinism_and_cache(folder):
    dev = torch.device('cpu')
    h1, h2 = FakeHist(), FakeHist()
    a = FolderData(folder, h1, 3, 16, dev, seed=1)
    b = FolderData(folder, h2, 3, 16, dev, seed=1, workers=1, prefetch=1)
    for _ in range(8):
        x, y = next(a), next(b)
        assert x['images'].shape == (3, 3, 16, 16) and x['histograms'].shape == (3, 3, 4, 4)
        assert float(x['images'].min()) >= 0.0 and float(x['images'].max()) <= 1.0
        assert torch.equal(x['images'], y['images']) and torch.equal(x['histograms'], y['histograms'])
    # 8 images in the folder: every target histogram is computed exactly once, the rest are cache hits
    assert h1.calls == len(a.cache) <= 8 and a.hits + a.misses == 8 * 3 * 2 and a.misses == h1.calls
    c = FolderData(folder, FakeHist(), 3, 16, dev, seed=1, cache_hists=False)
    assert torch.equal(next(c)['histograms'], next(FolderData(folder, FakeHist(), 3, 16, dev, seed=1))['histograms'])


def test_modes(folder):
    dev = torch.device('cpu')
    t = next(FolderData(folder, FakeHist(), 2, 16, dev, seed=2, test=True))
    assert set(t) == {'histograms'} and t['histograms'].shape == (2, 3, 4, 4)
    own = FolderData(folder, FakeHist(), 4, 16, dev, seed=3, hist_sampling=False)
    batch = next(own)
    # own histogram: equals the stand-in applied to the full-resolution image of the same index
    idx = np.random.RandomState(3).randint(0, len(own.paths), 4)
    for k, i in enumerate(idx):
        full = _load_rgb(own.paths[i]).unsqueeze(0)
        assert torch.allclose(batch['histograms'][k], FakeHist()(full)[0])
    with pytest.raises(FileNotFoundError):
        FolderData(folder + '/nothing_here', FakeHist(), 1, 16, dev)
    rgba = next(FolderData(folder, FakeHistAny(), 2, 16, dev, transparent=True))
    assert rgba['images'].shape == (2, 4, 16, 16) and float(rgba['images'][:, 3].min()) == 1.0   # opaque sources


def test_resize_crop_flip_and_grid(folder, tmp_path):
    p = sorted(os.listdir(folder))[1]
    full = _load_rgb(os.path.join(folder, p))
    small = _load_rgb(os.path.join(folder, p), 16)
    assert full.shape[0] == 3 and small.shape == (3, 16, 16)
    assert torch.equal(_load_rgb(os.path.join(folder, p), 16, flip=True), torch.flip(small, dims=(2,)))
    assert _load_rgb(os.path.join(folder, 'grey.png')).shape[0] == 3          # expand_greyscale
    out = os.path.join(tmp_path, 'grid.png')
    save_image_grid(torch.rand(5, 3, 8, 8), out, nrow=4)
    from PIL import Image
    assert Image.open(out).size == (4 * 10 + 2, 2 * 10 + 2)


def test_dataset_aug_prob_random_resized_crop(folder):
    """`dataset_aug_prob` (reference histoGAN/histoGAN.py:273-283): RandomResizedCrop(scale (0.5,1), ratio (0.98,1.02))
    with that probability, CenterCrop otherwise; deterministic for a seed; the crop box obeys the scale / ratio ranges."""
    from histogan_amd.data import _random_resized_crop_box
    rs = np.random.RandomState(1)
    for _ in range(200):
        w, h = int(rs.randint(16, 300)), int(rs.randint(16, 300))
        l, t, cw, ch = _random_resized_crop_box(w, h, rs)
        assert 0 <= l and 0 <= t and l + cw <= w and t + ch <= h and cw > 0 and ch > 0
    # on a square image the draw always fits: area in [0.5, 1] of the image, aspect in [0.98, 1.02] (up to rounding)
    for _ in range(100):
        l, t, cw, ch = _random_resized_crop_box(256, 256, rs)
        assert 0.49 <= cw * ch / 65536 <= 1.0 and 0.96 <= cw / ch <= 1.04
    dev = torch.device('cpu')
    a = FolderData(folder, FakeHist(), 4, 16, dev, seed=5, aug_prob=1.0, workers=2)
    b = FolderData(folder, FakeHist(), 4, 16, dev, seed=5, aug_prob=1.0, workers=2)
    c = FolderData(folder, FakeHist(), 4, 16, dev, seed=5, aug_prob=0.0, workers=2)
    xa, xb, xc = next(a)['images'], next(b)['images'], next(c)['images']
    assert xa.shape == (4, 3, 16, 16) and torch.equal(xa, xb)
    assert not torch.equal(xa, xc)                     # the crop changed what the batch shows
    assert float(xa.min()) >= 0.0 and float(xa.max()) <= 1.0


def test_histogram_cache_is_bounded_in_bytes(folder):
    dev = torch.device('cpu')
    one = 3 * 4 * 4 * 4                                 # bytes of one FakeHist histogram
    src = FolderData(folder, FakeHist(), 4, 16, dev, seed=0, workers=2, max_cache_bytes=3 * one)
    for _ in range(6):
        next(src)
    assert len(src.cache) == 3 and src.cache_bytes == 3 * one


# ---- parity with the reference Dataset's image transform (golden: tests/golden/make_golden_dataset.py) -------------
def test_item_images_equal_the_reference_dataset_transform():
    """`Dataset.__getitem__(i)['images']` of the UNMODIFIED reference class (histoGAN/histoGAN.py:270-303: Resize(256) with
    torchvision's truncating size arithmetic, CenterCrop with its half-to-even offsets, ToTensor) on the shipped target
    images incl. the 799 x 533 one: `_load_rgb` reproduces every pixel exactly."""
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'dataset.npz'))
    S = int(g['ds_meta'][0])
    for i, name in enumerate(g['ds_paths']):
        img = _load_rgb(os.path.join(GOLDEN_DIR, 'dataset_images', str(name)), S)
        want = torch.from_numpy(g[f'ds_item{i}_images_u8']).float().div(255)
        assert img.shape == want.shape and torch.equal(img, want), name


def test_folder_order_matches_the_golden_index():
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'dataset.npz'))
    fd = FolderData(os.path.join(GOLDEN_DIR, 'dataset_images'), FakeHist(), 2, 64, torch.device('cpu'))
    assert [p.name for p in fd.paths] == [str(n) for n in g['ds_paths']]
