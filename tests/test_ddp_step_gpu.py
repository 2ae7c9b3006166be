"""Data-parallel train step on hardware: two ranks (gloo, both on GPU 0) each run ONE
Trainer.train() step on half of a batch; the rank-averaged flat gradient buffers must equal those of a single process
on the whole batch -- the discriminator's exactly (up to summation order), the generator side's with the Hellinger
term in its global-batch form (one scalar all-reduce; histogan_amd/hist.py::_GlobalHellinger)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import relmax

pytestmark = pytest.mark.gpu

S_, CAP, HB, BG = 32, 4, 16, 4          # image size, capacity, histogram bins, GLOBAL batch


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


class _SliceRng:
    """Replays pre-drawn GLOBAL-batch latents / noise; a rank sees its slice (same draw order as the reference)."""

    def __init__(self, device, sl, L, LAT, seed):
        g = torch.Generator().manual_seed(seed)
        self.dev, self.sl = device, sl
        self.z = [torch.randn(BG, LAT, generator=g) for _ in range(4)]
        self.img_noise = [torch.rand(BG, S_, S_, 1, generator=g) for _ in range(2)]
        self.pl = torch.randn(BG, L - 2, LAT, generator=g)
        self.zi = self.ni = 0

    def noise(self, n, d):
        z = self.z[self.zi][self.sl]; self.zi += 1
        return z.to(self.dev)

    def noise_list(self, n, layers, d):
        return [(self.noise(n, d), layers)]

    def mixed_list(self, n, layers, d):
        return self.noise_list(n, 2, d) + self.noise_list(n, layers - 2, d)

    def image_noise(self, n, s):
        x = self.img_noise[self.ni][self.sl]; self.ni += 1
        return x.to(self.dev)

    def randn_like(self, t):
        return self.pl[self.sl].to(self.dev)


def _global_data():
    from oracle import rgbuv_hist as OH
    gen = torch.Generator().manual_seed(5)
    out = []
    for _ in range(2):
        img = torch.rand(BG, 3, S_, S_, generator=gen)
        hist = OH.rgbuv_hist(torch.rand(BG, 3, S_, S_, generator=gen), h=HB)
        out.append({'images': img, 'histograms': hist})
    return out


def _run_step(sl, step_no, tmp, tag):
    """One train() step on the samples `sl` of the global batch; returns (D flat grad, G flat grad, losses)."""
    from histoGAN import Trainer
    dev = torch.device('cuda', 0)
    torch.manual_seed(11)                                   # identical initial weights everywhere
    n = len(range(*sl.indices(BG)))
    tr = Trainer(tag, os.path.join(tmp, 'r'), os.path.join(tmp, 'm'), S_, CAP, batch_size=n, lr=2e-4, hist_bin=HB,
                 hist_insz=150, hist_resizing='interpolation', mixed_prob=1.1)
    tr.graph_mode = '0'
    tr.run_evaluate = tr.run_save = False
    tr.init_GAN()
    GAN = tr.GAN
    torch.manual_seed(12)
    with torch.no_grad():
        for blk in GAN.G.blocks:
            blk.to_noise1.weight.normal_(std=0.3); blk.to_noise2.weight.normal_(std=0.3)
    tr.loader = iter([{k: v[sl].to(dev) for k, v in b.items()} for b in _global_data()])
    tr.rng = _SliceRng(dev, sl, GAN.G.num_layers, GAN.G.latent_dim, 77)
    tr.steps = step_no
    tr.pl_mean = 0.05
    tr.train(alpha=2.0)
    torch.cuda.synchronize()
    return (GAN._flat_d.grad.cpu().numpy(), GAN._flat_g.grad.cpu().numpy(),
            np.array([tr.d_loss, tr.g_loss, tr.h_loss, tr.last_gp_loss], dtype=np.float64))


def _worker(rank, world, port, step_no, tmp, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        per = BG // world
        gd, gg, losses = _run_step(slice(rank * per, (rank + 1) * per), step_no, tmp, f'w{rank}')
        q.put((rank, gd, gg, losses))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, 'ERR', repr(e) + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('step_no', [1, 4, 0])     # a plain step, a gradient-penalty step, the GP + path-length step
# (step 0: the path-length term's `w_styles.std(dim=0)` is taken over the GLOBAL batch -- ddp.batch_std -- so two ranks on
# halves of the batch reproduce the single process on the whole of it)
def test_two_rank_step_equals_single_process(step_no, gpu_device, tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, step_no, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert all(not isinstance(r[1], str) for r in res), [r[2] for r in res if isinstance(r[1], str)]
    # both ranks hold the same (averaged) gradients and report the same (averaged / global) losses
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.allclose(res[0][3], res[1][3], rtol=0, atol=1e-7)
    gd1, gg1, l1 = _run_step(slice(0, BG), step_no, str(tmp_path), 'single')
    assert relmax(res[0][1], gd1) <= 2e-5, 'discriminator gradients'
    assert relmax(res[0][2], gg1) <= 2e-4, 'generator / S / H gradients (global Hellinger)'
    assert np.max(np.abs(res[0][3] - l1)) <= 1e-4 * max(1.0, float(np.max(np.abs(l1))))
