"""HistoGAN container + Trainer on the MI355X path.

API mirror of histoGAN/histoGAN.py (reference): `HistoGAN` (:634-715) and `Trainer` (:718-1139) keep
their constructor arguments, attribute names (`GAN.{S,H,G,D,SE,HE,GE,G_opt,D_opt}`, `steps`, ...),
method names and the checkpoint format (`torch.save(GAN.state_dict())`, `.config.json`).
`Trainer.train(alpha)` performs the same D step + G step (hinge loss, gradient penalty every 4th step,
Hellinger histogram loss, image-space path-length regulariser every 32nd step, DiffGrad, EMA schedule,
NaN recovery) -- restructured for the hardware:

* one process per GPU; gradients averaged over ranks (histogan_amd/ddp.py); the D-gradient all-reduce
  overlaps the generator forward of the G phase;
* flat parameter/gradient buffers, fused DiffGrad / EMA kernels (histogan_amd/optim.py);
* the D phase runs the generator under no_grad (the reference builds and discards that graph, :904-910);
* ONE device->host read-back per step (losses + NaN flag together) instead of >= 7 `.item()` syncs;
* latents / noise drawn on the device (rng='device'); rng='reference' reproduces the reference's CPU
  draws (`torch.randn(...).cuda()`, :166-189) for parity tests;
* the read-back is DEFERRED by one step (`lazy_stats`): step n's statistics travel to pinned host memory
  asynchronously and are looked at while step n+1 is already queued, so the GPU queue never drains at a step
  boundary.  Reading `d_loss` / `g_loss` / `h_loss` / `last_gp_loss` / `q_loss` / `pl_mean` (or print_log, save,
  evaluate) flushes what is pending first, so callers always see the values of the last finished step.

Provenance: `_device_step`, `_graphed_step`, `train`, the statistics plumbing and the data sources are original.
The API-compatibility shell -- `EMA`, `default`, `cast_list`, `is_empty`, `gradient_penalty`, `evaluate`,
`generate_truncated`, `print_log`, `save` / `load` / `clear`, `model_name`, `init_folders`, `config` handling and the
`Trainer.__init__` attribute block -- follows the reference method for method (histoGAN/histoGAN.py:107-163,
718-851, 1022-1139, `.cuda()` -> `.to(self.device)`), because the reference CLI, its checkpoints and its scripts
address exactly these names; none of it is on the timed path.
"""
import json
import os
import sys
from math import floor, log2
from pathlib import Path
from random import random
from shutil import rmtree
from time import perf_counter

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import grad as torch_grad

from . import ddp
from .augment import AugWrapper
from .conv import drain_pack_streams, enable_pack_cache, input_grads_only, prepack_async, weights_changed
from .hist import hellinger_loss
from .nets import Discriminator, Generator, HistVectorizer, StyleVectorizer
from .optim import DiffGrad, FlatParams, conv_first, ema_update

EPS = 1e-8
EXTS = ['jpg', 'png']
G_OVERLAP_DDP = os.environ.get('HG_G_OVERLAP_DDP', 'auto')   # the same under data parallelism: auto | 1 | 0 (see _device_step)
G_OVERLAP = os.environ.get('HG_G_OVERLAP', '1') != '0'   # G-phase generator forward on a second stream beside the D phase
G_STREAM_PRIO = int(os.environ.get('HG_G_STREAM_PRIO', '0'))   # HIP priority of that stream (0 normal, -1 high)
H_SIDE = os.environ.get('HG_H_SIDE', '1') != '0'               # histogram vectorizer on a second stream beside S
H_SIDE_GRAD = os.environ.get('HG_H_SIDE_GRAD', '1') != '0'     # ... in the grad-enabled G-phase forward too (its backward then runs there as well)
BATCH_S = os.environ.get('HG_BATCH_S', '1') != '0'             # both latent batches of a mixed draw through S at once
D_STEP_EARLY = os.environ.get('HG_D_STEP_EARLY', '1') != '0'   # D's optimizer step before main waits for that stream
# Plain steps replayed from a captured hipGraph (single process): HG_GRAPH = auto (default) | 1 | 0 | 2.
#   auto: decided from the first eager plain steps -- the graph is used when the host's enqueue work IS the step (small
#         batches, slow hosts: batch 4 at 256^2 runs 30 -> 20 ms; the enqueue time then equals the GPU-side step time,
#         ratio ~1); a GPU-bound step stays eager, where the side-stream overlaps (conv.py, `_g_stream`) are worth 2-4 %
#         that a replayed multi-branch graph does not keep (C3: 48.3 eager vs 50.1 ms replayed at a ratio of 0.5).  The
#         smallest ratio seen decides: the first steps of a process enqueue slowly (allocator growth, cold caches) and
#         say nothing about the steady state.
#   2:    the static-input step WITHOUT capture (A/B of the replay: identical parameter checksums, tools/graph_probe.py)
GRAPH_MODE = os.environ.get('HG_GRAPH', 'auto')
GRAPH_AUTO_RATIO = float(os.environ.get('HG_GRAPH_AUTO_RATIO', '0.9'))
GRAPH_GP = os.environ.get('HG_GRAPH_GP', '1') != '0'      # gradient-penalty steps replay from their own graph too
# G's convolution weights updated + re-packed under the tail of its backward (DiffGrad.step_early).  Measured at C3
# (profiles/r06_ab_early_gopt.json): 865.7 images/s with it, 867.7 without -- the HBM-bound update beside the latency-bound
# mapping-network backward slows that chain by what it saves at the step boundary; off by default.
EARLY_GOPT = os.environ.get('HG_EARLY_GOPT', '0') != '0'
# Data parallelism: the all-reduce of G's convolution-weight gradients (83 of 99.8 M parameters at C3) starts when the
# generator's fused backward node returns, under the rest of the backward (ddp.GradAllReduce.start_early)
EARLY_GREDUCE = os.environ.get('HG_EARLY_GREDUCE', '1') != '0'
LAZY_STATS = os.environ.get('HG_LAZY_STATS', '1') != '0'  # statistics of step n read while step n+1 is queued (0: blocking read-back every step)


def _lazy_field(name):
    """A statistic the reference keeps as a plain attribute (`self.d_loss = ...`): reading it first flushes the deferred
    read-backs, so the value is that of the last finished step."""
    key = '_stat_' + name

    def get(self):
        if self.__dict__.get('_pending'):
            self._drain(0, in_train=False)
        try:
            return self.__dict__[key]
        except KeyError:
            raise AttributeError(name) from None

    def set_(self, v):
        self.__dict__[key] = v
    return property(get, set_)


def _cat_batches(a, b):
    """torch.cat((a, b), 0) for two batches that carry no graph: two contiguous device copies into one buffer (aten's
    CatArrayBatchedCopy takes 185 us for the 2 x 25 MB of the C3 [fake; real] batch, 0.5 TB/s; the copies ~20 us)."""
    if a.requires_grad or b.requires_grad or a.shape[1:] != b.shape[1:] or a.dtype != b.dtype:
        return torch.cat((a, b), dim=0)
    out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
    out[:a.shape[0]].copy_(a)
    out[a.shape[0]:].copy_(b)
    return out


class NanException(Exception):
    pass


class EMA():
    def __init__(self, beta):
        super().__init__()
        self.beta = beta

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new


def _freeze_gc_once(owner):
    """The networks, optimizers and data source are ~10^5 long-lived Python objects; every full (generation-2) pass
    of the cyclic garbage collector walks all of them -- measured 50 ms, once every few steps, on a 62 ms step.  After
    the first step everything that will live for the whole run exists: collect once, then move the survivors to the
    permanent generation so later collections only look at young objects."""
    if not getattr(owner, '_gc_frozen', False):
        import gc
        gc.collect()
        gc.freeze()
        owner._gc_frozen = True


def default(value, d):
    return d if value is None else value


def cast_list(el):
    return el if isinstance(el, list) else [el]


def is_empty(t):
    if isinstance(t, torch.Tensor):
        return t.nelement() == 0
    return t is None


def set_requires_grad(model, bool):
    for p in model.parameters():
        p.requires_grad = bool


def gradient_penalty(images, output, weight=10):
    """10 * mean((||grad_x D(x)||_2 - 1)^2) on real images, double-differentiable (reference :156-163)."""
    batch_size = images.shape[0]
    with input_grads_only():    # d output / d images only: no conv node computes an (unused) weight gradient
        gradients = torch_grad(outputs=output, inputs=images, grad_outputs=torch.ones_like(output),
                               create_graph=True, retain_graph=True, only_inputs=True)[0]
    gradients = gradients.reshape(batch_size, -1)
    return weight * ((gradients.norm(2, dim=1) - 1) ** 2).mean()


class _Rng:
    """Latent / noise draws (reference :166-189).  'device': on the GPU; 'reference': CPU draw + copy."""

    def __init__(self, device, mode='device'):
        self.device, self.mode = device, mode

    def noise(self, n, latent_dim):
        if self.mode == 'reference':
            return torch.randn(n, latent_dim).to(self.device)
        return torch.randn(n, latent_dim, device=self.device)

    def noise_list(self, n, layers, latent_dim):
        return [(self.noise(n, latent_dim), layers)]

    def mixed_list(self, n, layers, latent_dim):
        tt = int(torch.rand(()).numpy() * layers)
        return self.noise_list(n, tt, latent_dim) + self.noise_list(n, layers - tt, latent_dim)

    def image_noise(self, n, im_size):
        if self.mode == 'reference':
            return torch.FloatTensor(n, im_size, im_size, 1).uniform_(0.0, 1.0).to(self.device)
        return torch.rand(n, im_size, im_size, 1, device=self.device)

    def randn_like(self, t):
        if self.mode == 'reference':
            return torch.randn(t.shape).to(self.device)
        return torch.randn_like(t)


def latent_to_w(style_vectorizer, latent_descr):
    zs = [z for z, _ in latent_descr]
    if BATCH_S and len(zs) > 1 and zs[0].is_cuda and all(z.shape == zs[0].shape for z in zs):
        # mixed latents (90 % of the steps): the two z batches through the 8-layer mapping network as ONE batch -- rows are
        # independent, and the 16 launches it saves are serial ~15 us ones at the head of every generator forward
        ws = style_vectorizer(torch.cat(zs, dim=0)).split(zs[0].shape[0], dim=0)
        return [(w, num_layers) for w, (_, num_layers) in zip(ws, latent_descr)]
    return [(style_vectorizer(z), num_layers) for z, num_layers in latent_descr]


def styles_def_to_tensor(styles_def):
    return torch.cat([t[:, None, :].expand(-1, n, -1) for t, n in styles_def], dim=1)


def evaluate_in_chunks(max_batch_size, model, *args):
    split_args = list(zip(*list(map(lambda x: x.split(max_batch_size, dim=0), args))))
    chunked_outputs = [model(*i) for i in split_args]
    if len(chunked_outputs) == 1:
        return chunked_outputs[0]
    return torch.cat(chunked_outputs, dim=0)


class HistoGAN(nn.Module):
    def __init__(self, image_size, latent_dim=512, style_depth=8, network_capacity=16, transparent=False,
                 fp16=False, steps=1, lr=1e-4, fq_layers=[], fq_dict_size=256, attn_layers=[], aug=False,
                 hist=64, device=None):
        super().__init__()
        if fp16:
            raise NotImplementedError('fp16/apex is not offered: the MI355X path is fp32 (as the reference default)')
        self.lr = lr
        self.aug = aug
        self.steps = steps
        self.ema_updater = EMA(0.995)
        self.S = StyleVectorizer(latent_dim, style_depth)
        self.H = HistVectorizer(hist, latent_dim, int(style_depth))
        self.G = Generator(image_size, latent_dim, network_capacity, transparent=transparent)
        self.D = Discriminator(image_size, network_capacity, fq_layers=fq_layers, fq_dict_size=fq_dict_size,
                               attn_layers=attn_layers, transparent=transparent)
        self.SE = StyleVectorizer(latent_dim, style_depth)
        self.HE = HistVectorizer(hist, latent_dim, int(style_depth))
        self.GE = Generator(image_size, latent_dim, network_capacity, transparent=transparent)
        # wrapper augmenting all images going into the discriminator (reference :658-662)
        self.D_aug = AugWrapper(self.D) if aug else None
        set_requires_grad(self.SE, False)
        set_requires_grad(self.HE, False)
        set_requires_grad(self.GE, False)
        self._init_weights()
        self.reset_parameter_averaging()

        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.to(device)
        # flat storage (after the move): generator side in the reference's optimizer order G, S, H (:668-670)
        # (convolution weights first: DiffGrad.step_early updates that region while the mapping networks' backward still runs)
        self._flat_g = FlatParams(conv_first(list(self.G.parameters()) + list(self.S.parameters()) + list(self.H.parameters())))
        self._flat_d = FlatParams(self.D.parameters())
        self._flat_ema = FlatParams(conv_first(list(self.GE.parameters()) + list(self.SE.parameters()) +
                                               list(self.HE.parameters())), with_grad=False)
        self.G_opt = DiffGrad(self._flat_g, lr=self.lr, betas=(0.5, 0.9))
        self.D_opt = DiffGrad(self._flat_d, lr=self.lr, betas=(0.5, 0.9))
        # replicas start identical under data parallelism
        for f in (self._flat_g, self._flat_d, self._flat_ema):
            ddp.broadcast_flat(f)
        ddp.broadcast_buffers(self)
        self._reduce_g = ddp.GradAllReduce(self._flat_g, ddp.BUCKETS)
        self._reduce_d = ddp.GradAllReduce(self._flat_d, ddp.BUCKETS)
        # packed conv weights are reused between the forward passes of one step (invalidated by the optimizers)
        enable_pack_cache(list(self.G.parameters()) + list(self.D.parameters()) + list(self.GE.parameters()))

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')
        for block in self.G.blocks:
            nn.init.zeros_(block.to_noise1.weight)
            nn.init.zeros_(block.to_noise2.weight)
            nn.init.zeros_(block.to_noise1.bias)
            nn.init.zeros_(block.to_noise2.bias)

    def EMA(self):
        ema_update(self._flat_ema, self._flat_g, self.ema_updater.beta)

    def reset_parameter_averaging(self):
        self.SE.load_state_dict(self.S.state_dict())
        self.HE.load_state_dict(self.H.state_dict())
        self.GE.load_state_dict(self.G.state_dict())

    def forward(self, x):
        return x


class SyntheticData:
    """Resident synthetic batches: images ~ U[0,1), target histograms = RGB-uv histograms of other random
    images (sum 1, all bins > 0) -- SURVEY.md section 8d.  Replaces the reference's DataLoader for benchmarks."""

    def __init__(self, hist_block, batch_size, image_size, device, pool=4, seed=0, channels=3):
        g = torch.Generator(device='cpu').manual_seed(seed)
        self.batches = []
        for _ in range(pool):
            img = torch.rand(batch_size, channels, image_size, image_size, generator=g).to(device)
            with torch.no_grad():
                hist = hist_block(torch.rand(batch_size, 3, image_size, image_size, generator=g).to(device))
            self.batches.append({'images': img, 'histograms': hist})
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return b


class Trainer():
    d_loss, g_loss, h_loss = _lazy_field('d_loss'), _lazy_field('g_loss'), _lazy_field('h_loss')
    last_gp_loss, q_loss, pl_mean = _lazy_field('last_gp_loss'), _lazy_field('q_loss'), _lazy_field('pl_mean')

    def __init__(self, name, results_dir, models_dir, image_size, network_capacity, transparent=False,
                 batch_size=4, mixed_prob=0.9, gradient_accumulate_every=1, lr=2e-4, num_workers=None,
                 save_every=1000, trunc_psi=0.6, fp16=False, fq_layers=[], fq_dict_size=256, attn_layers=[],
                 hist_method='inverse-quadratic', hist_resizing='sampling', hist_sigma=0.02, hist_bin=64,
                 hist_insz=150, aug_prob=0.0, dataset_aug_prob=0.0, aug_types=None, rng='device',
                 *args, **kwargs):
        from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
        if aug_types is None:
            aug_types = ['translation', 'cutout']
        self.GAN_params = [args, kwargs]
        self.GAN = None
        self.hist_method = hist_method
        self.hist_resizing = hist_resizing
        self.hist_sigma = hist_sigma
        self.hist_bin = hist_bin
        self.hist_insz = hist_insz
        self.histBlock = RGBuvHistBlock(insz=self.hist_insz, h=self.hist_bin, method=self.hist_method,
                                        resizing=self.hist_resizing, sigma=self.hist_sigma)
        self.name = name
        self.results_dir = Path(results_dir)
        self.models_dir = Path(models_dir)
        self.config_path = self.models_dir / name / '.config.json'
        assert log2(image_size).is_integer(), 'image size must be a power of 2 (64, 128, 256, 512, 1024)'
        self.image_size = image_size
        self.network_capacity = network_capacity
        self.transparent = transparent
        self.fq_layers = cast_list(fq_layers)
        self.fq_dict_size = fq_dict_size
        self.attn_layers = cast_list(attn_layers)
        self.aug_prob = aug_prob
        self.aug_types = aug_types
        self.dataset_aug_prob = dataset_aug_prob
        self.lr = lr
        self.batch_size = batch_size
        self.num_workers = num_workers
        self.mixed_prob = mixed_prob
        self.save_every = save_every
        self.steps = 0
        self.av = None
        self.trunc_psi = trunc_psi
        self.pl_mean = 0
        self.gradient_accumulate_every = gradient_accumulate_every
        assert not fp16, 'fp16 is not offered on the MI355X path (fp32 only)'
        self.fp16 = fp16
        self.d_loss = 0
        self.g_loss = 0
        self.last_gp_loss = 0
        self.last_cr_loss = 0
        self.q_loss = 0
        self.pl_length_ma = EMA(0.99)
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.rng = _Rng(self.device, rng)
        self.graph_mode = GRAPH_MODE          # 'auto' | '1' | '0' | '2' (see GRAPH_MODE above)
        self.lazy_stats = LAZY_STATS          # deferred read-back (module docstring); False: one blocking read-back per step
        self.keep_step_events = False         # bench.py: keep one timing event per step (GPU-side per-step durations)
        self.step_events = []
        self._pending = []                    # [(event, pinned stats, meta)] of steps whose statistics were not read yet
        self._nan_checkpoint = None           # NaN seen outside train(): the next train() call restores and raises
        self.is_main = ddp.rank() == 0
        self.run_evaluate = True      # benchmarks switch evaluate()/save() off (excluded from the metric)
        self.run_save = True
        self.init_folders()
        self.loader = None
        self.loader_evaluate = None

    def init_GAN(self):
        # a captured train-step graph points into the previous model's buffers: drop it (load() -> load_config() rebuilds
        # the GAN, e.g. on NaN recovery); the next eligible step captures again
        for k in ('_graphs', '_graph', '_graph_pool', '_gs', '_gs_first'):
            self.__dict__.pop(k, None)
        self.__dict__.setdefault('_pending', []).clear()
        self.__dict__.pop('_last_step_ev', None)
        args, kwargs = self.GAN_params
        self.GAN = HistoGAN(lr=self.lr, image_size=self.image_size, network_capacity=self.network_capacity,
                            transparent=self.transparent, fq_layers=self.fq_layers,
                            fq_dict_size=self.fq_dict_size, attn_layers=self.attn_layers, fp16=self.fp16,
                            hist=self.hist_bin, aug=self.aug_prob > 0, *args, **kwargs)

    def write_config(self):
        self.config_path.write_text(json.dumps(self.config()))

    def load_config(self):
        config = self.config() if not self.config_path.exists() else json.loads(self.config_path.read_text())
        self.image_size = config['image_size']
        self.network_capacity = config['network_capacity']
        self.transparent = config['transparent']
        self.fq_layers = config['fq_layers']
        self.fq_dict_size = config['fq_dict_size']
        self.attn_layers = config.pop('attn_layers', [])
        del self.GAN
        self.init_GAN()

    def config(self):
        return {'image_size': self.image_size, 'network_capacity': self.network_capacity,
                'transparent': self.transparent, 'fq_layers': self.fq_layers,
                'fq_dict_size': self.fq_dict_size, 'attn_layers': self.attn_layers}

    def set_synthetic_data_src(self, pool=4, seed=None):
        seed = ddp.rank() if seed is None else seed
        ch = 4 if self.transparent else 3
        self.loader = SyntheticData(self.histBlock, self.batch_size, self.image_size, self.device, pool, seed, ch)
        self.loader_evaluate = SyntheticData(self.histBlock, 4, min(self.image_size, 150), self.device, 1, seed + 977, ch)

    def set_data_src(self, folder):
        from .data import FolderData
        self.loader = FolderData(folder, self.histBlock, self.batch_size, self.image_size, self.device,
                                 transparent=self.transparent, seed=ddp.rank(), aug_prob=self.dataset_aug_prob)
        self.loader_evaluate = FolderData(folder, self.histBlock, 4, self.image_size, self.device,
                                          transparent=self.transparent, seed=977 + ddp.rank(), test=True)

    # ------------------------------------------------------------------------------------------
    def _g_stream(self):
        if getattr(self, '_gstream', None) is None:
            self._gstream = torch.cuda.Stream(device=self.device, priority=G_STREAM_PRIO)
        return self._gstream

    def _w_and_hw(self, style, hist_batch):
        GAN = self.GAN
        if (H_SIDE and hist_batch.is_cuda
                and ((H_SIDE_GRAD and self.__dict__.get('_extra_streams_ok', True)) or not torch.is_grad_enabled())
                and not torch.cuda.is_current_stream_capturing()):
            # the histogram vectorizer beside the mapping network instead of behind it -- two independent chains of 8 small
            # serial GEMMs at the head of the generator forward; with autograd recording, the engine runs each chain's
            # backward on the stream of its forward: the two chains are side by side at the tail of the backward as well
            from .nets import aux_stream
            main, aux = torch.cuda.current_stream(hist_batch.device), aux_stream(hist_batch.device)
            aux.wait_event(main.record_event())
            with torch.cuda.stream(aux):
                h_w_space = torch.unsqueeze(GAN.H(hist_batch), dim=1)
                h_w_space = torch.cat((h_w_space, h_w_space), dim=1)
            hist_batch.record_stream(aux)
            w_space = styles_def_to_tensor(latent_to_w(GAN.S, style))
            main.wait_stream(aux)
            h_w_space.record_stream(main)
            return w_space, h_w_space
        w_space = latent_to_w(GAN.S, style)
        h_w_space = GAN.H(hist_batch)
        h_w_space = torch.unsqueeze(h_w_space, dim=1)
        h_w_space = torch.cat((h_w_space, h_w_space), dim=1)
        return styles_def_to_tensor(w_space), h_w_space

    # ------------------------------------------------------------------------------------------
    # hipGraph replay of the plain step (no gradient penalty, no path-length term: 23 of every 32 steps)
    def _graph_eligible(self, apply_gradient_penalty, apply_path_penalty):
        """The step's ~1 800 launches cost the host ~45 ms of enqueue work; a captured graph replays them with one call.
        Captured: the plain step of a single-process run without DiffAugment / feature quantisation / gradient
        accumulation (those draw host-side random tables, keep batch-dependent buffers or change the launch sequence).
        HG_GRAPH=0 switches it off."""
        if self.graph_mode == '0' or apply_path_penalty or ddp.is_dist():    # (path-length steps branch on the host: eager)
            return False
        if apply_gradient_penalty and not GRAPH_GP:
            return False
        if self.gradient_accumulate_every != 1 or self.aug_prob > 0.0 or self.rng.mode != 'device':
            return False
        if any(q is not None for q in self.GAN.D.quantize_blocks) or getattr(self, '_graph_failed', False):
            return False
        if self.steps < getattr(self, '_graph_from', 6):          # a few eager steps first: caches, workspaces, gc freeze
            return False
        if self.graph_mode == 'auto':
            use = getattr(self, '_graph_auto', None)
            if use is None:
                r = getattr(self, '_host_ratio', [])
                if len(r) < 2:
                    return False
                use = self._graph_auto = min(r) > GRAPH_AUTO_RATIO
                if self.is_main:
                    print(f'train step: HG_GRAPH=auto -> {"hipGraph replay" if use else "eager"} (host enqueue / GPU step '
                          f'time of the first plain steps: {", ".join(f"{v:.2f}" for v in r)}; threshold {GRAPH_AUTO_RATIO})',
                          file=sys.stderr)
            return use
        return True

    def _graph_inputs(self):
        """Static buffers the captured step reads: the D phase's batch, the G phase's target histograms, and the two
        style-mixing split points (layers styled by the first latent; all of them = no mixing)."""
        gs = getattr(self, '_gs', None)
        if gs is None:
            b = next(self.loader)
            dev = self.device
            gs = self._gs = dict(images=torch.empty_like(b['images']), hist_d=torch.empty_like(b['histograms']),
                                 hist_g=torch.empty_like(b['histograms']),
                                 tt_d=torch.zeros((), dtype=torch.int64, device=dev),
                                 tt_g=torch.zeros((), dtype=torch.int64, device=dev))
            self._gs_first = b
        return gs

    def _fill_graph_inputs(self, gs):
        layers = self.GAN.G.num_layers - 2
        b = self.__dict__.pop('_gs_first', None) or next(self.loader)
        gs['images'].copy_(b['images'], non_blocking=True)
        gs['hist_d'].copy_(b['histograms'], non_blocking=True)
        gs['hist_g'].copy_(next(self.loader)['histograms'], non_blocking=True)
        mixed = random() < self.mixed_prob         # one choice per step (reference :891), a fresh split point per phase (:936)
        for key in ('tt_d', 'tt_g'):
            gs[key].fill_(int(torch.rand(()).numpy() * layers) if mixed else layers)

    def _latents_static(self, tt_dev, batch_size, layers, latent_dim):
        """Style tensor (B, layers, 512) with a DEVICE-side split point: layers < tt from latent 1, the rest from latent 2
        (`styles_def_to_tensor(latent_to_w(S, mixed_list(...)))` of the reference, :166-189, with the launch sequence
        independent of the draw)."""
        z1, z2 = self.rng.noise(batch_size, latent_dim), self.rng.noise(batch_size, latent_dim)
        if BATCH_S:
            w1, w2 = self.GAN.S(torch.cat((z1, z2), dim=0)).split(batch_size, dim=0)
        else:
            w1, w2 = self.GAN.S(z1), self.GAN.S(z2)
        first = (torch.arange(layers, device=self.device) < tt_dev).view(1, layers, 1)
        return torch.where(first, w1[:, None, :], w2[:, None, :])

    def _graphed_step(self, alpha, gp=False):
        """One step from a captured graph; gp: the gradient-penalty variant (its own graph, same memory pool -- the two
        are never in flight together)."""
        GAN = self.GAN
        drain_pack_streams()     # operands packed asynchronously by an eager step: ordered before the capture / replay
        gs = self._graph_inputs()
        self._fill_graph_inputs(gs)
        GAN.D_opt.prepare_replay()
        GAN.G_opt.prepare_replay()
        if self.graph_mode == '2':             # HG_GRAPH=2: the static-input step WITHOUT capture (A/B of the replay)
            for o in (GAN.D_opt, GAN.G_opt):
                o.graph_mode = True
            stats = self._device_step(alpha, gp, False, gs)
            for o in (GAN.D_opt, GAN.G_opt):
                o.graph_mode = False
            return stats
        graphs = self.__dict__.setdefault('_graphs', {})
        # alpha (the Hellinger weight) and the batch size are host scalars baked into the captured launches: the cache is
        # keyed on them, a call with another alpha captures its own graph instead of silently replaying the old value
        key = (bool(gp), float(alpha), int(self.batch_size))
        if key not in graphs:
            weights_changed()                  # every packed operand the step reads must be produced INSIDE the graph
            for o in (GAN.D_opt, GAN.G_opt):
                o.graph_mode = True
            graph = torch.cuda.CUDAGraph()
            try:
                from .conv import build_pack_plans
                build_pack_plans(self.device)
                torch.cuda.synchronize()
                pool = getattr(self, '_graph_pool', None)
                with torch.cuda.graph(graph, pool=pool):
                    stats = self._device_step(alpha, gp, False, gs)
                self._graph_pool = graph.pool()
            except Exception as e:             # capture refused (driver / library limitation): stay eager
                for o in (GAN.D_opt, GAN.G_opt):
                    o.graph_mode = False
                    o.step_count -= 1
                self._graph_failed = True
                torch.cuda.synchronize()
                print(f'hipGraph capture of the train step failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
                self._reset_after_failed_step()
                return self._device_step(alpha, gp, False, None)
            for o in (GAN.D_opt, GAN.G_opt):
                o.graph_mode = False
            graphs[key] = (graph, stats)
            self._graph = graph
            if not getattr(self, '_graph_logged', False) and self.is_main:
                self._graph_logged = True
                print(f'train step: replaying captured hipGraphs (HG_GRAPH={self.graph_mode}; latent / noise draws follow '
                      f'the static-input order -- for seed-reproducible runs pin HG_GRAPH=0 or 1)', file=sys.stderr)
        graph, stats = graphs[key]
        graph.replay()
        weights_changed()                      # eager steps in between must not trust operands packed by the graph
        return stats

    def _reset_after_failed_step(self):
        """State a partially executed `_device_step` can leave behind (an exception between the G phase's
        `set_requires_grad(D, False)` and its restore, half-written flat gradient buffers): put everything back so that
        the step can be run again from the top."""
        GAN = self.GAN
        set_requires_grad(GAN.D, True)
        for o in (GAN.D_opt, GAN.G_opt):
            o.zero_grad()
            o.flat.direct_ok = False
        GAN._reduce_d.finish()
        GAN._reduce_g.finish()
        weights_changed()

    def _device_step(self, alpha, apply_gradient_penalty, apply_path_penalty, gs=None):
        """All device work of one optimisation step (reference :853-989), no host synchronisation.  gs: static input
        buffers when the step is being captured as a hipGraph (None: eager).  Returns the stacked statistics
        [D loss, G loss, histogram loss, gradient penalty, quantize loss, path length]."""
        GAN = self.GAN
        dev = self.device
        zero = lambda: torch.zeros((), device=dev)
        mark = self._mark
        mark('start')
        total_disc_loss, total_gen_loss, total_hist_loss = zero(), zero(), zero()
        gp_val, q_val, pl_len = zero(), zero(), None

        batch_size = self.batch_size
        image_size = GAN.G.image_size
        latent_dim = GAN.G.latent_dim
        num_layers = GAN.G.num_layers
        Disc = GAN.D
        acc = self.gradient_accumulate_every
        has_vq = any(q is not None for q in Disc.quantize_blocks)
        # DiffAugment of everything the discriminator sees (reference :873-878, 905-908, 950-951)
        aug = (lambda im, detach=False: GAN.D_aug.augment(im, prob=self.aug_prob, types=self.aug_types, detach=detach)
               ) if self.aug_prob > 0.0 else (lambda im, detach=False: im)

        def w_and_hw_static(tt_key, hist_batch):
            w_styles = self._latents_static(gs[tt_key], batch_size, num_layers - 2, latent_dim)
            h_w_space = torch.unsqueeze(GAN.H(hist_batch), dim=1)
            return w_styles, torch.cat((h_w_space, h_w_space), dim=1)

        def g_forward():
            """latents, noise, target histograms and the generator forward of the G phase (reference :937-949)"""
            if gs is None:
                style = get_latents_fn(batch_size, num_layers - 2, latent_dim)
                noise = self.rng.image_noise(batch_size, image_size)
                hist_batch = next(self.loader)['histograms'].to(dev)
                w_styles, h_w_space = self._w_and_hw(style, hist_batch)
            else:
                hist_batch = gs['hist_g']
                w_styles, h_w_space = w_and_hw_static('tt_g', hist_batch)
                noise = self.rng.image_noise(batch_size, image_size)
            return noise, hist_batch, w_styles, h_w_space, GAN.G(w_styles, h_w_space, noise)

        # Under data parallelism (HG_G_OVERLAP_DDP = auto) the second-stream forward is used whenever every rank owns its GPU.
        # Nothing it touches depends on the D-gradient all-reduce (RCCL's own stream; waited for by D's optimizer step only):
        # one rank's view of a node (RCCL, world size 1 forced) runs 39.6 ms per step with it, 40.0 without.  It is switched
        # off when ranks SHARE a device -- the two-gloo-ranks-on-one-GPU test set-up ran 2.8-4.2 s per step with it and 0.26 s
        # without, and that follows the hardware-queue count the two processes put on the one GPU, not the step: with
        # GPU_MAX_HW_QUEUES=2 per process the same step takes 0.27 s (profiles/r05_ddp_overlap_shared_gpu.json).
        if not ddp.is_dist():
            ddp_ok = True
        elif G_OVERLAP_DDP in ('0', '1'):
            ddp_ok = G_OVERLAP_DDP == '1'
        else:
            ddp_ok = not ddp.ranks_share_a_device(dev)        # (collective on first use: every rank is in its first step here)
        overlap_g = G_OVERLAP and acc == 1 and ddp_ok
        self._extra_streams_ok = ddp_ok       # (ranks sharing a GPU: no further streams either -- H beside S under autograd, early all-reduce)
        if overlap_g and not getattr(self, '_warn_off', False):
            # parameters live on the default stream, part of the graph now runs on another one: the engine's stream
            # hand-over is intended
            fn = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if fn is not None:
                fn(False)
            self._warn_off = True
        early = None

        # ---- discriminator phase (reference :889-932)
        GAN.D_opt.zero_grad()
        for i in range(acc):
            if gs is None:
                get_latents_fn = self.rng.mixed_list if random() < self.mixed_prob else self.rng.noise_list
                style = get_latents_fn(batch_size, num_layers - 2, latent_dim)
                noise = self.rng.image_noise(batch_size, image_size)
                batch = next(self.loader)
                # d D(real) / d images is only needed by the gradient penalty (the reference sets requires_grad always, :897)
                image_batch = batch['images'].to(dev).detach().requires_grad_(apply_gradient_penalty)
                hist_batch = batch['histograms'].to(dev)
            else:
                get_latents_fn = None
                image_batch, hist_batch = gs['images'], gs['hist_d']
                if apply_gradient_penalty:
                    image_batch = image_batch.detach().requires_grad_(True)     # same storage: d D(real) / d images
            with torch.no_grad():   # the reference detaches this output; no graph is needed
                if gs is None:
                    w_styles, h_w_space = self._w_and_hw(style, hist_batch)
                else:
                    w_styles, h_w_space = w_and_hw_static('tt_d', hist_batch)
                    noise = self.rng.image_noise(batch_size, image_size)
                generated_images = GAN.G(w_styles, h_w_space, noise)
            mark('d_phase_g_forward')
            if overlap_g:
                # The generator forward of the G phase depends on nothing the D phase produces (same generator
                # weights, own latents): it runs on a second stream beside the discriminator's forward / backward --
                # two kernel mixes that stall on different resources (measured 20.4 -> 18.9 ms, tools/overlap_probe2.py).
                # Ordered after the forward above (which also packed this step's generator weights).
                main = torch.cuda.current_stream(dev)
                side2 = self._g_stream()
                side2.wait_event(main.record_event())
                with torch.cuda.stream(side2):
                    early = g_forward()
            # two passes, reference order: with feature quantisation (the codebook's moving averages and its loss depend
            # on the batch) and on gradient-penalty steps (the penalty's double backward then covers the real half only,
            # not a [fake; real] batch whose fake half carries zero gradients)
            if has_vq or apply_gradient_penalty:
                fake_output, fake_q_loss = Disc(aug(generated_images, True))
                real_output, real_q_loss = Disc(aug(image_batch))
            else:
                # one discriminator pass over [fake; real] (samples are independent: same values as two passes,
                # :911-912, but twice the pixels per launch on the small maps)
                both_output, both_q_loss = Disc(_cat_batches(aug(generated_images, True), aug(image_batch)))
                fake_output, real_output = both_output[:batch_size], both_output[batch_size:]
                fake_q_loss = real_q_loss = both_q_loss * 0.5
            mark('d_forward')
            divergence = (F.relu(1 + real_output) + F.relu(1 - fake_output)).mean()
            quantize_loss = (fake_q_loss + real_q_loss).mean()
            q_val = quantize_loss.detach()
            disc_loss = divergence + quantize_loss
            if apply_gradient_penalty:
                gp = gradient_penalty(image_batch, real_output)
                gp_val = gp.detach()
                disc_loss = disc_loss + gp
            disc_loss = disc_loss / acc
            disc_loss.backward()
            mark('d_backward')
            total_disc_loss += divergence.detach() / acc
        GAN._reduce_d.start()          # async all-reduce of D grads; overlaps the G forward below

        # ---- generator phase (reference :934-989)
        GAN.G_opt.zero_grad()
        # The reference lets the G-phase backward deposit gradients in D as well and throws them away at the
        # next D_opt.zero_grad() (:889); not computing them is the same result without D's weight-gradient pass.
        set_requires_grad(Disc, False)
        try:
            total_gen_loss, total_hist_loss, pl_len = self._g_phase(
                alpha, apply_path_penalty, gs, early, g_forward, aug, total_gen_loss, total_hist_loss)
        finally:
            set_requires_grad(Disc, True)
        if ddp.is_dist():
            GAN._reduce_g.start()
            GAN.G_opt.step_buckets(GAN._reduce_g)
        else:
            GAN._reduce_g()
            GAN.G_opt.step()
        if not self.__dict__.pop('_early_packed', False):
            prepack_async(GAN._flat_g.data)    # next step's generator operands, under the head of its forward
        mark('g_optimizer')

        return torch.stack([total_disc_loss, total_gen_loss, total_hist_loss, gp_val.reshape(()),
                            q_val.reshape(()), pl_len if pl_len is not None else zero()]).double()

    def _g_phase(self, alpha, apply_path_penalty, gs, early, g_forward, aug, total_gen_loss, total_hist_loss):
        """Generator phase of the step (reference :934-989); D's parameters are frozen by the caller."""
        GAN, dev, Disc, acc = self.GAN, self.device, self.GAN.D, self.gradient_accumulate_every
        pl_len = None
        d_updated = False
        def update_d():         # D must be updated before it scores the new fakes (reference order)
            if ddp.is_dist():   # bucket by bucket: the update of bucket i under the all-reduce of bucket i+1
                GAN.D_opt.step_buckets(GAN._reduce_d)
            else:
                GAN._reduce_d.finish()
                GAN.D_opt.step()

        if early is not None and D_STEP_EARLY:
            # the update only needs D's gradients: enqueued before this stream waits for the second one, it runs under
            # the tail of the G-phase generator forward when that is still in flight
            update_d()
            d_updated = True
        for i in range(acc):
            if early is not None:
                torch.cuda.current_stream(dev).wait_stream(self._g_stream())
                for t in early:
                    t.record_stream(torch.cuda.current_stream(dev))    # produced on the second stream, read on this one
                noise, hist_batch, w_styles, h_w_space, generated_images = early
                early = None
            else:
                noise, hist_batch, w_styles, h_w_space, generated_images = g_forward()
            if not d_updated:
                update_d()
                d_updated = True
            self._mark('g_forward_joined_d_updated')
            fake_output, _ = Disc(aug(generated_images))
            self._mark('g_phase_d_forward')
            generated_histograms = self.histBlock(generated_images, pre_relu=True)   # == histBlock(F.relu(.)), reference :955
            histogram_loss = hellinger_loss(hist_batch, generated_histograms, alpha)
            loss = fake_output.mean()
            gen_loss = loss + histogram_loss
            if apply_path_penalty:
                # (under data parallelism the statistic is over the GLOBAL batch, as in the single-GPU reference: ddp.batch_std)
                std = 0.1 / (ddp.batch_std(w_styles) + EPS)
                w_styles_2 = w_styles + self.rng.randn_like(w_styles) / (std + EPS)
                pl_images = GAN.G(w_styles_2, h_w_space, noise)
                pl_lengths = ((pl_images - generated_images) ** 2).mean(dim=(1, 2, 3))
                pl_len = pl_lengths.detach().mean()
                if not is_empty(self.pl_mean):
                    pl_loss = ((pl_lengths - self.pl_mean) ** 2).mean()
                    # reference :974: added only if not NaN.  A masked value would still send 0 * NaN gradients back through
                    # the second generator pass, so this is a real branch -- one host sync on every 32nd step.
                    if not bool(torch.isnan(pl_loss)):
                        gen_loss = gen_loss + pl_loss
            gen_loss = gen_loss / acc
            self._mark('g_phase_d_forward_hist_loss')
            # at the end of the generator's fused backward node: start the all-reduce of G's convolution-weight gradients
            # (data parallelism) or -- single process, opt-in -- update them
            early_opt = ((EARLY_GREDUCE and self.__dict__.get('_extra_streams_ok', True) if ddp.is_dist() else EARLY_GOPT)
                         and acc == 1 and not apply_path_penalty
                         and not torch.cuda.is_current_stream_capturing())
            if early_opt:
                from . import gfused
                gfused.AFTER_BLOCKS = self._early_g_update
            try:
                gen_loss.backward()
            finally:
                if early_opt:
                    gfused.AFTER_BLOCKS = None
            self._mark('g_backward')
            total_gen_loss = total_gen_loss + loss.detach() / acc
            total_hist_loss = total_hist_loss + histogram_loss.detach() / acc
        return total_gen_loss, total_hist_loss, pl_len

    def train(self, alpha=2):
        assert self.loader is not None, ('You must first initialize the data source with '
                                         '`.set_data_src(<folder of images>)` or `.set_synthetic_data_src()`')
        torch.autograd.set_detect_anomaly(False)
        if self.GAN is None:
            self.init_GAN()
        if self._nan_checkpoint is not None:       # a NaN surfaced while a statistic was read between two train() calls
            self._raise_nan(self._nan_checkpoint)
        GAN = self.GAN
        GAN.train()
        _freeze_gc_once(self)
        t_host0 = perf_counter()
        apply_gradient_penalty = self.steps % 4 == 0
        apply_path_penalty = self.steps % 32 == 0
        graphed = self._graph_eligible(apply_gradient_penalty, apply_path_penalty)
        try:
            if graphed:
                stats = self._graphed_step(alpha, apply_gradient_penalty)
            else:
                stats = self._device_step(alpha, apply_gradient_penalty, apply_path_penalty, None)
        except NanException:
            raise
        except Exception:
            if self.GAN is not None:
                self._reset_after_failed_step()
            raise

        # host time spent enqueueing this step (everything before the read-back); graphed: the replay call
        self.host_enqueue_ms = (perf_counter() - t_host0) * 1e3
        self.last_step_graphed = bool(graphed) and self.graph_mode != '2' and not getattr(self, '_graph_failed', False)
        self._t_host0 = t_host0

        # ---- ONE read-back for everything the host needs (reference: >= 7 `.item()` syncs), deferred by one step: the
        # packed statistics go to pinned host memory asynchronously; what the host waits for below is the PREVIOUS
        # step's copy, which completes while this step's launches are already queued -- the queue never drains at a step
        # boundary (profiles/r02_step_timeline.txt: 0.69 + 0.47 ms idle per boundary with the blocking read-back).
        if ddp.is_dist():
            # ONE small collective per step: the six statistics are averaged, the NaN flag is summed (any rank -> > 0), so
            # every rank raises / recovers together
            nan_flag = torch.isnan(stats[:4]).any().double().reshape(1)
            packed = torch.cat([stats, nan_flag * ddp.world_size()])
            torch.distributed.all_reduce(packed, op=torch.distributed.ReduceOp.SUM)
            packed /= ddp.world_size()
        else:
            packed = torch.cat([stats, torch.zeros(1, dtype=stats.dtype, device=stats.device)])
        host = self._host_buffer()
        host.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        checkpoint_num = floor(self.steps / self.save_every)
        self._pending.append((ev, host, dict(step=self.steps, gp=apply_gradient_penalty, pl=apply_path_penalty,
                                             checkpoint=checkpoint_num, host_ms=self.host_enqueue_ms,
                                             prev_ev=self.__dict__.get('_last_step_ev'),
                                             eager=not graphed)))
        self._last_step_ev = ev
        if self.keep_step_events:
            self.step_events.append((self.steps, ev))
        will_save = self.run_save and self.steps % self.save_every == 0
        will_eval = self.run_evaluate and (self.steps % 1000 == 0 or (self.steps % 100 == 0 and self.steps < 2500))
        # save / evaluate look at this step's result (the reference checks for NaN before it saves, :1002-1014)
        self._drain(1 if (self.lazy_stats and not will_save and not will_eval) else 0, in_train=True)

        # moving averages (reference :996-1000)
        if self.steps % 10 == 0 and self.steps > 20000:
            GAN.EMA()
        if self.steps <= 25000 and self.steps % 1000 == 2:
            GAN.reset_parameter_averaging()

        if will_save:
            self.save(checkpoint_num)
        if will_eval:
            self.evaluate(floor(self.steps / 1000))
        self.steps += 1
        self.av = None

    def _early_g_update(self):
        """Called by the generator's fused backward node when its last convolution weight gradient is enqueued: DiffGrad on the
        convolution-weight region of G's flat buffer and the re-pack of its operands run on a stream of their own beside the
        rest of the backward (mapping networks, style projections); `G_opt.step()` finishes the other parameters."""
        GAN = self.GAN
        st = self.__dict__.get('_opt_stream')
        if st is None:
            st = self._opt_stream = torch.cuda.Stream(device=self.device)
        if ddp.is_dist():
            GAN._reduce_g.start_early(st)
        elif GAN.G_opt.step_early(st):
            with torch.cuda.stream(st):
                self._early_packed = bool(prepack_async(GAN._flat_g.data))

    def _mark(self, name):
        """Phase marker of the step on the CURRENT stream (tools/phase_probe.py sets `phase_events = []`; None: no-op)."""
        pe = self.__dict__.get('phase_events')
        if pe is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            pe.append((name, ev, perf_counter()))

    # ---- deferred statistics -----------------------------------------------------------------
    def _host_buffer(self):
        """Pinned host staging for one step's packed statistics (two rotate: one in flight, one being read)."""
        ring = self.__dict__.setdefault('_host_ring', [])
        for buf in ring:
            if not any(buf is h for _, h, _ in self._pending):
                return buf
        buf = torch.empty(7, dtype=torch.float64).pin_memory()
        ring.append(buf)
        return buf

    def _drain(self, keep, in_train):
        """Consume the statistics of all pending steps but the newest `keep`: losses, the path-length moving average
        (reference :991-994) and the NaN check (:1002-1010).  in_train=False (a statistic read between two train()
        calls): a NaN is remembered and raised by the next train() call, inside the caller's retry loop."""
        while len(self._pending) > keep:
            ev, host_t, meta = self._pending.pop(0)
            ev.synchronize()
            host = host_t.numpy().copy()
            if self.graph_mode == 'auto' and self.__dict__.get('_graph_auto') is None and meta['eager'] \
                    and meta['step'] >= 2 and not (meta['gp'] or meta['pl']) and meta['prev_ev'] is not None:
                # HG_GRAPH=auto: the host's share of an eager plain step = its enqueue time over the GPU-side time between the
                # end-of-step events of this step and the previous one (host-bound: the two are equal; GPU-bound: the host
                # runs ahead).  Event times, not host wall times: with the deferred read-back the host's wall time of a
                # call says when the PREVIOUS step finished.
                gpu_ms = meta['prev_ev'].elapsed_time(ev)
                self.__dict__.setdefault('_host_ratio', []).append(meta['host_ms'] / max(gpu_ms, 1e-3))
            meta['prev_ev'] = None
            # generator loss incl. the histogram term and the gradient penalty, as the reference's checks on gen_loss /
            # disc_loss (:978, :925); under data parallelism the flag is the all-reduced one so every rank raises
            has_nan = bool(host[6] > 0 or np.isnan(host[:4]).any())
            d = self.__dict__
            d['_stat_d_loss'], d['_stat_g_loss'], d['_stat_h_loss'] = float(host[0]), float(host[1]), float(host[2])
            if meta['gp']:
                d['_stat_last_gp_loss'] = float(host[3])
            d['_stat_q_loss'] = float(host[4])
            if meta['pl'] and not np.isnan(host[5]):
                d['_stat_pl_mean'] = self.pl_length_ma.update_average(d.get('_stat_pl_mean', 0), float(host[5]))
            if has_nan:
                self._pending.clear()
                if in_train:
                    self._raise_nan(meta['checkpoint'])
                self._nan_checkpoint = meta['checkpoint']
                return

    def flush(self):
        """Consume every pending read-back NOW and apply the reference's NaN handling (:1002-1010) to what it finds: restore
        the last checkpoint and raise NanException.  The deferred read-back looks at step n's statistics inside the
        train() call of step n + 1; the LAST step of a run has no successor, so call this after a training loop and before
        weights leave the process -- `save()` does it itself (a NaN step is never written), `evaluate()` and `print_log()`
        read the statistics and thereby arm the same exception for the next `train()` / `flush()` / `save()`."""
        if self._pending:
            self._drain(0, in_train=False)
        if self._nan_checkpoint is not None:
            self._raise_nan(self._nan_checkpoint)

    finalize = flush

    def _raise_nan(self, checkpoint_num):
        # save from NaN errors (reference :1002-1010)
        self._nan_checkpoint = None
        self._pending.clear()
        print(f'NaN detected for generator or discriminator. Loading from checkpoint #{checkpoint_num}')
        self.load(checkpoint_num)
        raise NanException

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, num=0, hist_batch=None, num_image_tiles=4, latents=None, n=None, save_noise_latent=False,
                 load_noise_file=None, load_latent_file=None):
        self.GAN.eval()
        if hist_batch is None:
            batch = next(self.loader_evaluate)
            hist_batch = batch['histograms'].to(self.device)
        ext = 'jpg' if not self.transparent else 'png'
        num_rows = num_image_tiles
        if latents is None and n is None:
            latent_dim = self.GAN.G.latent_dim
            image_size = self.GAN.G.image_size
            num_layers = self.GAN.G.num_layers
            if load_noise_file is not None:
                n = torch.tensor(np.load(load_noise_file)).to(self.device)
            else:
                n = self.rng.image_noise(num_rows ** 2, image_size)
            if load_latent_file is not None:
                # [(latent (n, 512), layers)] as saved below: an object array (the reference's plain np.save / np.load of
                # this list, :1046, :1060, only round-trips on NumPy versions that still built ragged object arrays)
                latents = [(torch.as_tensor(np.asarray(z, dtype=np.float32)).to(self.device), int(layers))
                           for z, layers in np.load(load_latent_file, allow_pickle=True)]
            else:
                latents = self.rng.noise_list(num_rows ** 2, num_layers - 2, latent_dim)
        generated_images = self.generate_truncated(self.GAN.SE, self.GAN.HE, self.GAN.GE, hist_batch, latents, n,
                                                   trunc_psi=self.trunc_psi)
        if num is not None and self.is_main:
            from .data import save_image_grid
            save_image_grid(generated_images, str(self.results_dir / self.name / f'{str(num)}-ema.{ext}'),
                            nrow=num_rows)
        if save_noise_latent:
            Path(f'temp/{self.name}').mkdir(parents=True, exist_ok=True)
            np.save(f'temp/{self.name}/{str(num)}-noise.npy', n.clone().cpu().numpy())
            arr = np.empty((len(latents), 2), dtype=object)
            for i, (z, layers) in enumerate(latents):
                arr[i, 0], arr[i, 1] = z.detach().cpu().numpy(), int(layers)
            np.save(f'temp/{self.name}/{str(num)}-latents.npy', arr, allow_pickle=True)
        return generated_images

    @torch.no_grad()
    def generate_truncated(self, S, H, G, hist_batch, style, noi, trunc_psi=0.75):
        latent_dim = G.latent_dim
        if self.av is None:
            z = self.rng.noise(2000, latent_dim)
            samples = evaluate_in_chunks(self.batch_size, S, z).cpu().numpy()
            self.av = np.mean(samples, axis=0)
            self.av = np.expand_dims(self.av, axis=0)
        w_space = []
        for tensor, num_layers in style:
            tmp = S(tensor)
            av_torch = torch.from_numpy(self.av).to(self.device)
            tmp = trunc_psi * (tmp - av_torch) + av_torch
            w_space.append((tmp, num_layers))
        h_w_space = H(hist_batch)
        h_w_space = torch.unsqueeze(h_w_space, dim=1)
        h_w_space = torch.cat((h_w_space, h_w_space), dim=1)
        for i in range(int(np.log2(np.sqrt(w_space[0][0].shape[0])))):
            h_w_space = torch.cat((h_w_space, h_w_space), dim=0)
        w_styles = styles_def_to_tensor(w_space)
        generated_images = evaluate_in_chunks(self.batch_size, G, w_styles, h_w_space, noi)
        return generated_images.clamp_(0.0, 1.0)

    def print_log(self):
        if not self.is_main:
            return
        if hasattr(self, 'h_loss'):
            print(f'\nG: {self.g_loss:.2f} | H: {self.h_loss:.2f} | D: {self.d_loss:.2f} | GP: '
                  f'{self.last_gp_loss:.2f} | PL: {self.pl_mean:.2f} | CR: {self.last_cr_loss:.2f} | Q: '
                  f'{self.q_loss:.2f}')
        else:
            print(f'\nG: {self.g_loss:.2f} | D: {self.d_loss:.2f} | GP: {self.last_gp_loss:.2f} | PL: '
                  f'{self.pl_mean:.2f} | CR: {self.last_cr_loss:.2f} | Q: {self.q_loss:.2f}')

    def model_name(self, num):
        return str(self.models_dir / self.name / f'model_{num}.pt')

    def init_folders(self):
        (self.results_dir / self.name).mkdir(parents=True, exist_ok=True)
        (self.models_dir / self.name).mkdir(parents=True, exist_ok=True)

    def clear(self):
        rmtree(f'./models/{self.name}', True)
        rmtree(f'./results/{self.name}', True)
        rmtree(str(self.config_path), True)
        self.init_folders()

    def save(self, num):
        self.flush()           # never persist the weights of a step whose NaN check is still pending (raises NanException)
        if self.is_main:       # rank 0 alone writes (replicas are identical)
            torch.save(self.GAN.state_dict(), self.model_name(num))
            self.write_config()

    def load(self, num=-1):
        self.load_config()
        name = num
        if num == -1:
            file_paths = [p for p in Path(self.models_dir / self.name).glob('model_*.pt')]
            saved_nums = sorted(map(lambda x: int(x.stem.split('_')[1]), file_paths))
            if len(saved_nums) == 0:
                return
            name = saved_nums[-1]
            print(f'continuing from previous epoch - {name}')
        self.steps = name * self.save_every
        self.GAN.load_state_dict(torch.load(self.model_name(name), map_location=self.device))
        ddp.broadcast_buffers(self.GAN)
        weights_changed()
