"""SURVEY.md section 8 rows f-2 and f-4 against goldens of the UNMODIFIED reference (tests/golden/make_golden_dataset.py).

f-2  histogan_amd.data.FolderData vs the reference `Dataset.__getitem__` flow (histoGAN/histoGAN.py:253-307) on the
     shipped target images: per-image target histograms (GPU kernel vs the reference's CPU block) <= 1e-5, the item's
     `histograms` with the reference's recorded draws (two image indices + interpolation ratio) <= 1e-5 with the
     interpolation weights applied exactly, `images` bit-equal (tests/test_data_cpu.py covers that without a GPU).
f-4  `Trainer.generate_truncated` (:1064-1091) with fixed latents / noise / av vs the reference method's own output
     <= 1e-5; the generate-mode and train-mode call sequences of the root CLI (histoGAN.py:66-202) restated call by call.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, relmax

pytestmark = pytest.mark.gpu

IMG_DIR = os.path.join(GOLDEN_DIR, 'dataset_images')


@pytest.fixture(scope='module')
def g():
    z = np.load(os.path.join(GOLDEN_DIR, 'dataset.npz'))
    return {k: z[k] for k in z.files}


def _folder_data(g, dev, resizing='sampling', B=1):
    from histogan_amd.data import FolderData
    from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
    S, HB, INSZ = (int(v) for v in g['ds_meta'])
    blk = RGBuvHistBlock(insz=INSZ, h=HB, method='inverse-quadratic', resizing=resizing)
    return FolderData(IMG_DIR, blk, B, S, dev, seed=0), S, HB


# ---- f-2 --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('resizing,key', [('sampling', 'ds_own'), ('interpolation', 'ds_own_interp')])
def test_target_histograms_match_the_reference_dataset(g, resizing, key, gpu_device):
    """Test-mode items (`test=True`, :304-307: the image's own histogram through RGBuvHistBlock(device='cpu')) for the
    trainer-default 'sampling' resize (h x h pixels of the full-resolution image: 256^2, 799 x 533, 1024^2) and for
    'interpolation' (bilinear to 150 x 150)."""
    fd, S, HB = _folder_data(g, gpu_device, resizing)
    n = len(fd.paths)
    hists = fd._hist(range(n), {'full': {}})
    assert hists.shape == (n, 3, HB, HB)
    for i in range(n):
        assert relmax(hists[i].cpu().numpy(), g[f'{key}{i}']) <= 1e-5, (i, fd.paths[i].name)
    assert fd.misses == n and len(fd.cache) == n
    again = fd._hist(range(n), {'full': {}})                   # second pass: device cache only
    assert torch.equal(again, hists) and fd.hits == n


def test_training_items_match_the_reference_dataset(g, gpu_device):
    """Training-mode items (:293-303): 'images' = Resize + CenterCrop + ToTensor of image i, 'histograms' =
    hist_interpolation(hist(img1), hist(img2)) with img1, img2, ratio drawn per item.  FolderData draws from its own
    generator, so the reference's recorded draws are put into its plan; everything downstream (decode pool, GPU
    histograms, device cache, interpolation) is the product path."""
    fd, S, HB = _folder_data(g, gpu_device)
    n = len(fd.paths)
    for i in range(n):
        inds, ratio = g[f'ds_item{i}_inds'], g[f'ds_item{i}_ratio'].astype(np.float32)
        plan = fd._plan()
        plan['img'], plan['h1'], plan['h2'], plan['ratio'] = np.array([i]), inds[:1], inds[1:], ratio
        from histogan_amd.data import _load_rgb
        plan['small'] = [fd.pool.submit(_load_rgb, fd.paths[i], S, False, False, None)]
        fd.queue.clear(); fd.queue.append(plan)
        while len(fd.queue) < fd.prefetch:
            fd.queue.append(fd._plan())
        batch = next(fd)
        want_img = torch.from_numpy(g[f'ds_item{i}_images_u8']).float().div(255)
        assert torch.equal(batch['images'][0].cpu(), want_img)
        got = batch['histograms'][0].cpu().numpy()
        assert relmax(got, g[f'ds_item{i}_histograms']) <= 1e-5, i
        # the interpolation itself is exact: the same fp32 expression on the two cached histograms
        h1, h2 = fd.cache[int(inds[0])], fd.cache[int(inds[1])]
        r = torch.from_numpy(ratio).to(gpu_device)
        assert torch.equal(batch['histograms'][0], h1 * r + h2 * (1 - r))


# ---- f-4 --------------------------------------------------------------------------------------------------------------
def _small_trainer(g, dev, tmp_path, name='gt'):
    from histoGAN import Trainer
    S_, CAP, LAT, HBs, NT = (int(v) for v in g['gt_meta'])
    tr = Trainer(name, str(tmp_path / 'results'), str(tmp_path / 'models'), S_, CAP, batch_size=3, hist_bin=HBs,
                 hist_insz=150, hist_resizing='interpolation', latent_dim=LAT, style_depth=3)
    tr.run_evaluate = tr.run_save = False
    tr.init_GAN()
    sd = lambda p: {k[len(p) + 1:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith(p + '/')}
    for mods, p in (((tr.GAN.G, tr.GAN.GE), 'gt_G'), ((tr.GAN.S, tr.GAN.SE), 'gt_S'), ((tr.GAN.H, tr.GAN.HE), 'gt_H')):
        for m in mods:
            m.load_state_dict(sd(p))
    from histogan_amd.conv import weights_changed
    weights_changed()
    return tr, NT


@pytest.mark.parametrize('psi', [0.75, 0.4])
def test_generate_truncated_matches_the_reference_method(g, psi, gpu_device, tmp_path):
    """The reference `Trainer.generate_truncated` run unbound on the reference's own small networks (golden): truncation
    around the given `av`, the histogram embedding appended as the last two styles, `evaluate_in_chunks` over
    batch_size = 3 (4 images -> 3 + 1), clamp to [0, 1]."""
    tr, NT = _small_trainer(g, gpu_device, tmp_path)
    dev = gpu_device
    tr.av = g['gt_av'].copy()
    z, noi, hb = (torch.from_numpy(g[k]).to(dev) for k in ('gt_z', 'gt_noise', 'gt_hist'))
    L = tr.GAN.G.num_layers
    imgs = tr.generate_truncated(tr.GAN.SE, tr.GAN.HE, tr.GAN.GE, hb, [(z, L - 2)], noi, trunc_psi=psi)
    want = g[f'gt_images_psi{psi}']
    assert imgs.shape == want.shape
    assert float(np.abs(imgs.cpu().numpy() - want).max()) <= 1e-5
    # evaluate() with explicit latents / noise is the same call (reference :1022-1062) and writes the grid
    tr.trunc_psi = psi
    tr.av = g['gt_av'].copy()
    out = tr.evaluate('grid', hist_batch=hb, num_image_tiles=NT, latents=[(z, L - 2)], n=noi)
    assert torch.equal(out, imgs) and (tmp_path / 'results' / 'gt' / 'grid-ema.jpg').exists()


def _retry_call(f, fargs=(), tries=3, exceptions=Exception):
    """retry.api.retry_call as the CLI uses it (histoGAN.py:198)."""
    for t in range(tries):
        try:
            return f(*fargs)
        except exceptions:
            if t == tries - 1:
                raise


def test_cli_call_sequences(g, gpu_device, tmp_path, capsys, monkeypatch):
    """The call order of the root CLI's `train_from_folder` (histoGAN.py:66-202) restated against the drop-in Trainer with
    the CLI's keyword set: train mode (Trainer(**kwargs) -> clear / load -> set_data_src -> retry_call(train, [alpha]) ->
    print_log every 50th step) and generate mode for a .npy target histogram and for a .jpg / .png target image
    (RGBuvHistBlock(device=<ordinal>) -> ToTensor -> histblock -> duplicate to num_image_tiles^2 rows -> evaluate with
    save_noise_latent, then again with load_noise_file / load_latent_file: same images)."""
    from histoGAN import NanException, Trainer
    from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
    monkeypatch.chdir(tmp_path)                            # the CLI writes ./temp/<name>/ relative to the working directory
    kw = dict(batch_size=2, gradient_accumulate_every=1, image_size=32, network_capacity=2, transparent=False, lr=2e-4,
              num_workers=None, save_every=2, trunc_psi=0.75, fp16=False, fq_layers=[], fq_dict_size=256, attn_layers=[],
              hist_insz=150, hist_bin=16, hist_sigma=0.02, hist_resizing='sampling', hist_method='inverse-quadratic',
              aug_prob=0.0, dataset_aug_prob=0.0, aug_types=None)
    name, results, models = 'cli', str(tmp_path / 'results'), str(tmp_path / 'models')
    # ---- train mode, --new
    model = Trainer(name, results, models, **kw)
    model.clear()
    model.set_data_src(IMG_DIR)
    num_train_steps = 5
    for _ in range(num_train_steps - model.steps):
        _retry_call(model.train, fargs=[2], tries=3, exceptions=NanException)
        if _ % 50 == 0:
            model.print_log()
    assert model.steps == 5 and 'G:' in capsys.readouterr().out
    assert (tmp_path / 'models' / name / 'model_2.pt').exists()          # save_every = 2: checkpoints 0, 1, 2
    # ---- resume (not --new): load(-1) picks the newest checkpoint and the step counter
    model = Trainer(name, results, models, **kw)
    model.load(-1)
    assert model.steps == 4
    # ---- generate mode, .npy target (histoGAN.py:117-134)
    num_image_tiles = 3
    hist_file = str(tmp_path / 'target.npy')
    blk = RGBuvHistBlock(insz=150, h=16, resizing='sampling', method='inverse-quadratic', sigma=0.02,
                         device=torch.cuda.current_device())           # an ORDINAL, as the CLI passes it (:138)
    img = torch.from_numpy(g['ds_item1_images_u8']).float().div(255).unsqueeze(0).to(device=torch.cuda.current_device())
    np.save(hist_file, blk(img).cpu().numpy())
    h = torch.from_numpy(np.load(hist_file)).to(device=torch.cuda.current_device())
    num_image_tiles = num_image_tiles - num_image_tiles % 2
    for i in range(int(np.log2(num_image_tiles))):
        h = torch.cat((h, h), dim=0)
    a = model.evaluate('generated-target', hist_batch=h, num_image_tiles=num_image_tiles, save_noise_latent=True,
                       load_noise_file=None, load_latent_file=None)
    assert a.shape == (4, 3, 32, 32) and (tmp_path / 'results' / name / 'generated-target-ema.jpg').exists()
    noise_f, lat_f = f'temp/{name}/generated-target-noise.npy', f'temp/{name}/generated-target-latents.npy'
    assert os.path.exists(noise_f) and os.path.exists(lat_f)
    # ---- generate mode, image target (:135-155) + replay of the saved noise
    from PIL import Image
    pil = Image.open(os.path.join(IMG_DIR, str(g['ds_paths'][1])))
    img2 = torch.unsqueeze(torch.from_numpy(np.asarray(pil, dtype=np.float32) / 255.0).permute(2, 0, 1), dim=0).to(
        device=torch.cuda.current_device())
    h2 = blk(img2)
    for i in range(int(np.log2(num_image_tiles))):
        h2 = torch.cat((h2, h2), dim=0)
    b = model.evaluate('generated-img', hist_batch=h2, num_image_tiles=num_image_tiles, save_noise_latent=False,
                       load_noise_file=noise_f, load_latent_file=None)
    assert b.shape == a.shape and torch.isfinite(b).all()
    # the saved noise + latents reproduce the first grid (target_noise_file / target_latent_file of the CLI, :50-51)
    c = model.evaluate('generated-replay', hist_batch=h, num_image_tiles=num_image_tiles, save_noise_latent=False,
                       load_noise_file=noise_f, load_latent_file=lat_f)
    assert torch.equal(c, a)
