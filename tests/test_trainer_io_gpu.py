"""Trainer around the step: image-folder source with GPU target histograms, checkpoint
save/load with the reference's state_dict key names, evaluate() grid output (row f-4)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _make_folder(path, n=6, size=40):
    from PIL import Image
    rs = np.random.RandomState(0)
    os.makedirs(path, exist_ok=True)
    for i in range(n):
        arr = (rs.rand(size + i, size + 2 * i, 3) * 255).astype(np.uint8)       # different sizes, like a real folder
        Image.fromarray(arr).save(os.path.join(path, f'im{i}.png' if i % 2 else f'im{i}.jpg'))


def test_folder_source_checkpoint_and_evaluate(gpu_device, tmp_path):
    from histoGAN import Trainer
    folder = str(tmp_path / 'imgs')
    _make_folder(folder)
    kw = dict(batch_size=2, hist_bin=16, hist_insz=32, hist_resizing='interpolation', save_every=1000)
    tr = Trainer('io', str(tmp_path / 'results'), str(tmp_path / 'models'), 32, 2, **kw)
    tr.set_data_src(folder)
    batch = next(tr.loader)
    assert batch['images'].shape == (2, 3, 32, 32) and batch['images'].is_cuda
    assert batch['histograms'].shape == (2, 3, 16, 16)
    assert torch.allclose(batch['histograms'].sum(dim=(1, 2, 3)), torch.ones(2, device=gpu_device), atol=1e-4)
    tr.run_evaluate = False
    tr.train(alpha=2)                    # step 0: saves checkpoint 0 (steps % save_every == 0), GP + PL step
    tr.train(alpha=2)
    assert tr.steps == 2 and np.isfinite(tr.d_loss) and np.isfinite(tr.g_loss)
    ckpt = tmp_path / 'models' / 'io' / 'model_0.pt'
    assert ckpt.exists() and (tmp_path / 'models' / 'io' / '.config.json').exists()

    # checkpoint keys == the reference's HistoGAN.state_dict() layout (golden file holds G./D. names of the reference)
    sd = torch.load(str(ckpt), map_location='cpu')
    gold = np.load(os.path.join(GOLDEN_DIR, 'nets_small.npz'))
    ref_g = {k[2:] for k in gold.files if k.startswith('G/')}
    ref_d = {k[2:] for k in gold.files if k.startswith('D/')}
    mine_g = {k[2:] for k in sd if k.startswith('G.')}
    mine_d = {k[2:] for k in sd if k.startswith('D.')}
    # same parameter names per block (the golden nets are smaller: compare the name patterns of block 0 and the tails)
    pat = lambda names: {n for n in names if n.startswith('blocks.0.') or not n.startswith('blocks.')}
    assert pat(ref_g) == pat(mine_g)
    assert pat(ref_d) == pat(mine_d)
    assert {k.split('.')[0] for k in sd} >= {'S', 'H', 'G', 'D', 'SE', 'HE', 'GE'}

    # resume in a fresh trainer, same weights, then evaluate() writes the EMA sample grid
    tr2 = Trainer('io', str(tmp_path / 'results'), str(tmp_path / 'models'), 32, 2, **kw)
    tr2.load(-1)
    assert tr2.steps == 0
    for (k1, v1), (k2, v2) in zip(sorted(tr2.GAN.state_dict().items()), sorted(sd.items())):
        assert k1 == k2 and torch.equal(v1.cpu(), v2)
    tr2.set_data_src(folder)
    imgs = tr2.evaluate(num=7, num_image_tiles=2)
    assert imgs.shape == (4, 3, 32, 32) and float(imgs.min()) >= 0.0 and float(imgs.max()) <= 1.0
    assert (tmp_path / 'results' / 'io' / '7-ema.jpg').exists()


def test_transparent_rgba_steps(gpu_device, tmp_path):
    """transparent=True: 4-channel generator output / discriminator input (reference :374-376, 577); the histogram
    takes the first three channels (RGBuvHistBlock.py:98-99)."""
    from histoGAN import Trainer
    tr = Trainer('rgba', str(tmp_path / 'r'), str(tmp_path / 'm'), 32, 2, transparent=True, batch_size=2, hist_bin=16,
                 hist_insz=32, hist_resizing='interpolation')
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    tr.train(alpha=2)
    tr.train(alpha=2)
    assert np.isfinite(tr.d_loss) and np.isfinite(tr.g_loss) and np.isfinite(tr.h_loss)
    assert tr.GAN.G.blocks[-1].to_rgb.conv.weight.shape[0] == 4 and tr.GAN.D.blocks[0].conv_res.weight.shape[1] == 4
    assert tr.evaluate(num=None).shape[1] == 4


@pytest.mark.parametrize('lazy', [False, True])
def test_nan_recovery_raises_nanexception(lazy, gpu_device, tmp_path):
    """Reference error convention (histoGAN/histoGAN.py:1002-1010): a NaN loss reloads the last checkpoint and raises
    NanException from train().  With the deferred read-back (lazy_stats, the default) the exception comes from the NEXT
    train() call -- still inside the caller's `retry_call(model.train, exceptions=NanException)` loop -- or is armed as
    soon as a statistic is read; with lazy_stats = False from the same call, as in the reference."""
    from histoGAN import NanException, Trainer
    tr = Trainer('nan', str(tmp_path / 'r'), str(tmp_path / 'm'), 32, 2, batch_size=2, hist_bin=16, hist_insz=32,
                 hist_resizing='interpolation', save_every=1000)
    tr.run_evaluate = False
    tr.lazy_stats = lazy
    tr.set_synthetic_data_src()
    tr.train(alpha=2)                                    # step 0 writes checkpoint 0
    good = {k: v.clone() for k, v in tr.GAN.state_dict().items()}
    with torch.no_grad():
        tr.GAN.D.to_logit.weight.fill_(float('nan'))
    from histogan_amd.conv import weights_changed
    weights_changed()
    if lazy:
        tr.train(alpha=2)                                # the NaN step itself: its statistics are still in flight
        assert not np.isfinite(tr.d_loss)                # reading one flushes them and arms the exception
    with pytest.raises(NanException):
        tr.train(alpha=2)
    # the checkpoint was reloaded: finite weights again (those of step 0's save), training continues
    assert all(torch.isfinite(v).all() for v in tr.GAN.state_dict().values())
    assert tr.steps == 0
    tr.train(alpha=2)
    assert np.isfinite(tr.d_loss) and np.isfinite(tr.g_loss)
    del good


def test_trailing_nan_step_is_caught_by_flush_and_never_saved(gpu_device, tmp_path):
    """with the deferred read-back the LAST step of a run has no following train() call to look at its
    statistics.  Trainer.flush() (alias finalize()) drains them and applies the reference's NaN handling (:1002-1010);
    save() calls it first, so a direct save after the loop cannot persist NaN weights."""
    from histoGAN import NanException, Trainer
    from histogan_amd.conv import weights_changed
    tr = Trainer('nanlast', str(tmp_path / 'r'), str(tmp_path / 'm'), 32, 2, batch_size=2, hist_bin=16, hist_insz=32,
                 hist_resizing='interpolation', save_every=1000)
    tr.run_evaluate = False
    assert tr.lazy_stats
    tr.set_synthetic_data_src()
    tr.train(alpha=2)                                    # step 0 writes checkpoint 0
    with torch.no_grad():
        tr.GAN.D.to_logit.weight.fill_(float('nan'))
    weights_changed()
    tr.train(alpha=2)                                    # the run's last step: NaN, statistics still in flight
    with pytest.raises(NanException):
        tr.save(5)                                       # direct save after the loop: flushes, restores, raises
    assert not (tmp_path / 'm' / 'nanlast' / 'model_5.pt').exists()
    assert all(torch.isfinite(v).all() for v in tr.GAN.state_dict().values()) and tr.steps == 0
    tr.flush()                                           # nothing pending any more: no-op
    tr.train(alpha=2)
    with torch.no_grad():
        tr.GAN.D.to_logit.weight.fill_(float('nan'))
    weights_changed()
    tr.train(alpha=2)
    with pytest.raises(NanException):
        tr.finalize()


def test_gradient_accumulation_steps(gpu_device, tmp_path):
    """gradient_accumulate_every = 2 with mixed_prob = 0 (reference :889-932: losses divided by the count, gradients
    accumulated over the micro-batches before one optimizer step)."""
    from histoGAN import Trainer
    kw = dict(batch_size=2, hist_bin=16, hist_insz=32, hist_resizing='interpolation', mixed_prob=0.0)
    tr = Trainer('acc', str(tmp_path / 'r'), str(tmp_path / 'm'), 32, 2, gradient_accumulate_every=2, **kw)
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    for _ in range(3):
        tr.train(alpha=2)
    assert tr.steps == 3 and np.isfinite(tr.d_loss) and np.isfinite(tr.g_loss) and np.isfinite(tr.h_loss)
    # every parameter of G / S / H / D received a finite gradient through the flat buffers (the discriminator's may be
    # identically zero: at this initialisation the hinge can be inactive for every sample of a 2 x 2 batch)
    for flat in (tr.GAN._flat_g, tr.GAN._flat_d):
        assert torch.isfinite(flat.grad).all()
    assert float(tr.GAN._flat_g.grad.abs().sum()) > 0
