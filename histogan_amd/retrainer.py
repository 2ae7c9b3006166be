"""ReHistoGAN container + trainer on the MI355X path.

API mirror of ReHistoGAN/rehistoGAN.py (reference): `recoloringGAN` (:637-718) and `recoloringTrainer` (:721-1226)
keep their constructor arguments, attribute names (`GAN.{ED,H,G,D,G_opt,D_opt}`, `steps`, ...), method names and
the checkpoint format.  `recoloringTrainer.train(alpha, beta, gamma)` performs the reference's D step + G step
(:895-1073: hinge loss, gradient penalty every 4th step, gamma * adversarial + alpha * Hellinger histogram loss +
beta * L1/Sobel/Laplacian reconstruction loss [+ variance loss], DiffGrad, NaN recovery), restructured as the
HistoGAN trainer is (histogan_amd/trainer.py): flat parameter/gradient buffers with fused DiffGrad, the D phase's
encoder-decoder + head under no_grad (the reference builds and discards that graph), one `[fake; real]`
discriminator pass, no D weight gradients in the G phase, one read-back per step, RCCL all-reduce of the flat
gradient buffers under data parallelism.

Provenance: the step (`train`), `_recolor`, the loss plumbing and the data sources are original.  The API-compatibility
shell -- the `recoloringTrainer.__init__` attribute block, `config` / `write_config` / `load_config`, `print_log`,
`model_name`, `init_folders`, `clear`, `save`, `load` and `evaluate`'s parameter list -- follows the reference method for
method (ReHistoGAN/rehistoGAN.py:721-893, 1076-1226), because its CLI, checkpoints and scripts address these names;
none of it is on the timed path.
"""
import json
from math import floor, log2, pi
from pathlib import Path
from shutil import rmtree

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ddp, reops
from .conv import enable_pack_cache, weights_changed
from .hist import hellinger_loss
from .nets import Discriminator, HistVectorizer
from .optim import DiffGrad, FlatParams
from .renets import RecoloringEncoderDecoder, RecoloringGAN
from .trainer import (G_OVERLAP, NanException, SyntheticData, _freeze_gc_once, _Rng, cast_list, gradient_penalty,
                      set_requires_grad)

SOBEL_X = ((1, 0, -1), (2, 0, -2), (1, 0, -1))
SOBEL_Y = ((1, 2, 1), (0, 0, 0), (-1, -2, -1))
LAPLACIAN = ((0, 1, 0), (1, -4, 1), (0, 1, 0))


def get_gaussian_kernel(kernel_size=15, sigma=3, channels=3):
    """The reference's depthwise Gaussian (:207-225) as its (channels,1,k,k) weight tensor (all slices equal)."""
    x_coord = torch.arange(kernel_size)
    x_grid = x_coord.repeat(kernel_size).view(kernel_size, kernel_size)
    xy_grid = torch.stack([x_grid, x_grid.t()], dim=-1).float()
    mean = (kernel_size - 1) / 2.
    variance = sigma ** 2.
    k = (1. / (2. * pi * variance)) * torch.exp(-torch.sum((xy_grid - mean) ** 2., dim=-1) / (2 * variance))
    k = k / torch.sum(k)
    return k.view(1, 1, kernel_size, kernel_size).repeat(channels, 1, 1, 1)


def gaussian_op(x, kernel=None):
    if kernel is None:
        kernel = get_gaussian_kernel(kernel_size=15, sigma=15, channels=3)
    return reops.gaussian_valid(x, kernel)


def laplacian_op(x, kernel=None):
    return reops.stencil3(x, LAPLACIAN if kernel is None else kernel)


def sobel_op(x, dir=0, kernel=None):
    return reops.stencil3(x, (SOBEL_X if dir == 0 else SOBEL_Y) if kernel is None else kernel)


class reconstruction_loss(object):
    """'L1' | '1st gradient' (Sobel magnitude) | '2nd gradient' (Laplacian); reference :279-326."""

    def __init__(self, loss):
        self.loss = loss

    def compute_loss(self, input, target):
        if self.loss == 'L1':
            return torch.mean(torch.abs(input - target))
        if self.loss == '1st gradient':
            ig = torch.sqrt(sobel_op(input, 0) ** 2 + sobel_op(input, 1) ** 2)
            tg = torch.sqrt(sobel_op(target, 0) ** 2 + sobel_op(target, 1) ** 2)
            return torch.mean(torch.abs(ig - tg))
        if self.loss == '2nd gradient':
            # the stencil is linear: one pass over the difference image
            return torch.mean(torch.abs(laplacian_op(input - target)))
        return None


class recoloringGAN(nn.Module):
    def __init__(self, image_size, latent_dim=512, style_depth=8, network_capacity=16, transparent=False, fp16=False,
                 steps=1, lr=1e-4, fq_layers=[], fq_dict_size=256, attn_layers=[], hist=64, skip_conn_to_GAN=False,
                 fixed_gan_weights=False, initialize_gan=False, internal_hist=False, device=None):
        super().__init__()
        if fp16:
            raise NotImplementedError('fp16/apex is not offered: the MI355X path is fp32 (as the reference default)')
        self.lr = lr
        self.steps = steps
        self.fixed_gan_weights = fixed_gan_weights
        self.internal_hist = internal_hist
        self.skip_conn_to_GAN = skip_conn_to_GAN
        self.ED = RecoloringEncoderDecoder(image_size, network_capacity=network_capacity, hist=hist,
                                           latent_dim=latent_dim, style_depth=style_depth,
                                           skip_conn_to_GAN=skip_conn_to_GAN, internal_hist=internal_hist)
        self.H = HistVectorizer(hist, latent_dim, int(style_depth))
        self.G = RecoloringGAN(image_size, latent_dim, network_capacity, transparent=transparent)
        self.D = Discriminator(image_size, network_capacity, fq_layers=fq_layers, fq_dict_size=fq_dict_size,
                               attn_layers=attn_layers, transparent=transparent)
        self._init_weights(initializeGAN=initialize_gan)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.to(device)
        if fixed_gan_weights:
            # the reference leaves G and H out of the optimizer (:672-677); not computing their (unused) gradients
            # is the same update
            set_requires_grad(self.G, False)
            set_requires_grad(self.H, False)
            learnable = list(self.ED.parameters())
        else:
            learnable = list(self.ED.parameters()) + list(self.G.parameters()) + list(self.H.parameters())
        self._flat_g = FlatParams(learnable)
        self._flat_d = FlatParams(self.D.parameters())
        self.G_opt = DiffGrad(self._flat_g, lr=self.lr, betas=(0.5, 0.9))
        self.D_opt = DiffGrad(self._flat_d, lr=self.lr, betas=(0.5, 0.9))
        for f in (self._flat_g, self._flat_d):
            ddp.broadcast_flat(f)
        if fixed_gan_weights and ddp.is_dist():
            for p in list(self.G.parameters()) + list(self.H.parameters()):
                torch.distributed.broadcast(p.data, 0)
        self._reduce_g = ddp.GradAllReduce(self._flat_g)
        self._reduce_d = ddp.GradAllReduce(self._flat_d)
        enable_pack_cache(list(self.ED.parameters()) + list(self.G.parameters()) + list(self.D.parameters()))

    def _init_weights(self, initializeGAN=False):
        if initializeGAN:
            for block in self.G.blocks:
                nn.init.zeros_(block.to_noise1.weight)
                nn.init.zeros_(block.to_noise2.weight)
                nn.init.zeros_(block.to_noise1.bias)
                nn.init.zeros_(block.to_noise2.bias)
            for m in self.H.modules():
                if isinstance(m, (nn.Conv2d, nn.Linear)):
                    nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')
        for net in (self.ED, self.D):
            for m in net.modules():
                if isinstance(m, (nn.Conv2d, nn.Linear)):
                    nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')

    def forward(self, x):
        return x


class recoloringTrainer():
    def __init__(self, name, results_dir, models_dir, image_size, network_capacity, transparent=False, batch_size=4,
                 mixed_prob=0.9, gradient_accumulate_every=1, lr=2e-4, num_workers=None, save_every=1000,
                 trunc_psi=0.6, fp16=False, fq_layers=[], fq_dict_size=256, attn_layers=[],
                 hist_method='inverse-quadratic', hist_resizing='sampling', hist_sigma=0.02, hist_bin=64,
                 hist_insz=150, fixed_gan_weights=False, skip_conn_to_GAN=False, rec_loss='laplacian',
                 initialize_gan=False, variance_loss=True, internal_hist=False, change_hyperparameters=False,
                 change_hyperparameters_after=100000, rng='device', *args, **kwargs):
        from histogram_classes.RGBuvHistBlock import RGBuvHistBlock
        self.GAN_params = [args, kwargs]
        self.GAN = None
        self.hist_method = hist_method
        self.hist_resizing = hist_resizing
        self.hist_sigma = hist_sigma
        self.hist_bin = hist_bin
        self.change_hyperparameters_after = change_hyperparameters_after
        self.hist_insz = hist_insz
        self.rec_loss = rec_loss
        self.internal_hist = internal_hist
        self.change_hyperparameters = change_hyperparameters
        self.variance_loss = variance_loss
        self.fixed_gan_weights = fixed_gan_weights
        self.skip_conn_to_GAN = skip_conn_to_GAN
        self.initialize_gan = initialize_gan
        self.device = torch.device('cuda', torch.cuda.current_device())
        mk = lambda: RGBuvHistBlock(insz=self.hist_insz, h=self.hist_bin, method=self.hist_method,
                                    resizing=self.hist_resizing, sigma=self.hist_sigma)
        self.histBlock = mk()
        if variance_loss is True:
            self.histBlock_input = mk()
            self.gaussKernel = get_gaussian_kernel(kernel_size=15, sigma=5, channels=3).to(self.device)
        if self.rec_loss is None:
            self.rec_loss_func = reconstruction_loss('L1')
        elif self.rec_loss == 'sobel':
            self.rec_loss_func = reconstruction_loss('1st gradient')
        elif self.rec_loss == 'laplacian':
            self.rec_loss_func = reconstruction_loss('2nd gradient')
        else:
            raise Exception('Unknown reconstruction losst!')
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
        self.lr = lr
        self.batch_size = batch_size
        self.num_workers = num_workers
        self.mixed_prob = mixed_prob
        self.save_every = save_every
        self.steps = 0
        self.av = None
        self.trunc_psi = trunc_psi
        self.gradient_accumulate_every = gradient_accumulate_every
        assert not fp16, 'fp16 is not offered on the MI355X path (fp32 only)'
        self.fp16 = fp16
        self.d_loss = 0
        self.g_loss = 0
        self.last_gp_loss = 0
        self.last_cr_loss = 0
        self.q_loss = 0
        if self.variance_loss is True:
            self.var_loss = 0
        self.rng = _Rng(self.device, rng)
        self.is_main = ddp.rank() == 0
        self.run_evaluate = True
        self.run_save = True
        self.init_folders()
        self.loader = None
        self.loader_evaluate = None

    def init_GAN(self):
        args, kwargs = self.GAN_params
        self.GAN = recoloringGAN(lr=self.lr, image_size=self.image_size, network_capacity=self.network_capacity,
                                 transparent=self.transparent, fq_layers=self.fq_layers,
                                 fq_dict_size=self.fq_dict_size, attn_layers=self.attn_layers, fp16=self.fp16,
                                 hist=self.hist_bin, fixed_gan_weights=self.fixed_gan_weights,
                                 skip_conn_to_GAN=self.skip_conn_to_GAN, initialize_gan=self.initialize_gan,
                                 internal_hist=self.internal_hist, *args, **kwargs)

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
        self.loader = SyntheticData(self.histBlock, self.batch_size, self.image_size, self.device, pool, seed)
        self.loader_evaluate = SyntheticData(self.histBlock, 4, self.image_size, self.device, 1, seed + 977)

    def set_data_src(self, folder, sampling=True):
        from .data import FolderData
        self.loader = FolderData(folder, self.histBlock, self.batch_size, self.image_size, self.device,
                                 transparent=self.transparent, seed=ddp.rank(), hist_sampling=sampling,
                                 hflip=True)     # transforms.RandomHorizontalFlip(), ReHistoGAN/rehistoGAN.py:362
        self.loader_evaluate = FolderData(folder, self.histBlock, 4, self.image_size, self.device,
                                          transparent=self.transparent, seed=977 + ddp.rank(),
                                          hist_sampling=sampling)

    # ------------------------------------------------------------------------------------------
    def _g_stream(self):
        if getattr(self, '_gstream', None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
        return self._gstream

    def _recolor(self, image_batch, hist_batch, noise):
        """H -> encoder-decoder -> recolouring head, the four wirings of the reference (:934-952)."""
        GAN = self.GAN
        h_w_space = GAN.H(hist_batch)
        ed_hist = h_w_space if self.internal_hist else hist_batch
        if self.skip_conn_to_GAN:
            image_latent, rgb, latent_a, latent_b = GAN.ED(image_batch, ed_hist)
            return GAN.G(image_latent, rgb, h_w_space, noise, latent_a, latent_b)
        image_latent, rgb = GAN.ED(image_batch, ed_hist)
        return GAN.G(image_latent, rgb, h_w_space, noise)

    def train(self, alpha=32, beta=1.5, gamma=4):
        assert self.loader is not None, ('You must first initialize the data source with '
                                         '`.set_data_src(<folder of images>)` or `.set_synthetic_data_src()`')
        if self.steps >= self.change_hyperparameters_after and self.change_hyperparameters:
            # (the reference assigns self.alpha/gamma/beta here and keeps using the arguments, :900-904)
            self.alpha, self.gamma, self.beta = 8, 2, 1
        torch.autograd.set_detect_anomaly(False)
        if self.GAN is None:
            self.init_GAN()
        GAN = self.GAN
        GAN.train()
        _freeze_gc_once(self)
        dev = self.device
        zero = lambda: torch.zeros((), device=dev)
        total_disc_loss, total_gen_loss, total_rec_loss, total_hist_loss, total_var_loss = (zero() for _ in range(5))
        gp_val, q_val = zero(), zero()
        batch_size = self.batch_size
        image_size = GAN.G.image_size
        Disc = GAN.D
        acc = self.gradient_accumulate_every
        apply_gradient_penalty = self.steps % 4 == 0

        def g_forward():
            """batch, noise and the recolouring forward of the G phase (reference :971-990)"""
            batch = next(self.loader)
            image_batch = batch['images'].to(dev)
            hist_batch = batch['histograms'].to(dev)
            noise = self.rng.image_noise(batch_size, image_size)
            return image_batch, hist_batch, self._recolor(image_batch, hist_batch, noise)

        # the G phase's forward on a second stream beside the discriminator's forward / backward (single-GPU runs; see
        # histogan_amd/trainer.py)
        overlap_g = G_OVERLAP and acc == 1 and not ddp.is_dist()
        if overlap_g and not getattr(self, '_warn_off', False):
            fn = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if fn is not None:
                fn(False)
            self._warn_off = True
        early = None

        # ---- discriminator phase (reference :927-969)
        GAN.D_opt.zero_grad()
        for i in range(acc):
            batch = next(self.loader)
            # d D(real) / d images is only needed by the gradient penalty (the reference sets requires_grad always, :897)
            image_batch = batch['images'].to(dev).detach().requires_grad_(apply_gradient_penalty)
            hist_batch = batch['histograms'].to(dev)
            noise = self.rng.image_noise(batch_size, image_size)
            with torch.no_grad():       # the reference detaches this output; no graph is needed
                generated_images = self._recolor(image_batch, hist_batch, noise)
            if overlap_g:
                main = torch.cuda.current_stream(dev)
                side2 = self._g_stream()
                side2.wait_event(main.record_event())     # after the forward above (it packed this step's weights)
                with torch.cuda.stream(side2):
                    early = g_forward()
            # two passes with feature quantisation (batch-dependent codebook) and on gradient-penalty steps (the double
            # backward then covers the real half only)
            if apply_gradient_penalty or any(q is not None for q in Disc.quantize_blocks):
                fake_output, fake_q_loss = Disc(generated_images)
                real_output, real_q_loss = Disc(image_batch)
                quantize_loss = (fake_q_loss + real_q_loss).mean()
            else:
                both_output, both_q_loss = Disc(torch.cat((generated_images, image_batch), dim=0))
                fake_output, real_output = both_output[:batch_size], both_output[batch_size:]
                quantize_loss = both_q_loss.mean()
            divergence = (F.relu(1 + real_output) + F.relu(1 - fake_output)).mean()
            q_val = quantize_loss.detach()
            disc_loss = divergence + quantize_loss
            if apply_gradient_penalty:
                gp = gradient_penalty(image_batch, real_output)
                gp_val = gp.detach()
                disc_loss = disc_loss + gp
            disc_loss = disc_loss / acc
            disc_loss.backward()
            total_disc_loss += divergence.detach() / acc
        GAN._reduce_d.start()

        # ---- generator phase (reference :971-1048)
        GAN.G_opt.zero_grad()
        set_requires_grad(Disc, False)
        d_updated = False
        for i in range(acc):
            if early is not None:
                torch.cuda.current_stream(dev).wait_stream(self._g_stream())
                for t in early:
                    t.record_stream(torch.cuda.current_stream(dev))
                image_batch, hist_batch, generated_images = early
                early = None
            else:
                image_batch, hist_batch, generated_images = g_forward()
            if not d_updated:           # D is updated before it scores the new fakes (reference order)
                GAN._reduce_d.finish()
                GAN.D_opt.step()
                d_updated = True
            fake_output, _ = Disc(generated_images)
            d_loss = gamma * fake_output.mean()
            generated_histograms = self.histBlock(generated_images, pre_relu=True)   # == histBlock(F.relu(.)), reference :955
            histogram_loss = hellinger_loss(hist_batch, generated_histograms, alpha)
            rec_loss = beta * self.rec_loss_func.compute_loss(image_batch, generated_images)
            gen_loss = d_loss + histogram_loss + rec_loss
            if self.variance_loss is True:
                with torch.no_grad():   # the histogram of the target histogram "image" (:1023) is data
                    input_histograms = self.histBlock_input(F.relu(hist_batch))
                    hist_gap = torch.sum(torch.abs(hist_batch - input_histograms))
                    input_gauss = gaussian_op(image_batch, kernel=self.gaussKernel)
                    input_spread = torch.std(torch.std(input_gauss, dim=2), dim=2)
                generated_gauss = gaussian_op(generated_images, kernel=self.gaussKernel)
                var_loss = -1 * (beta / 10) * hist_gap * torch.mean(torch.abs(
                    input_spread - torch.std(torch.std(generated_gauss, dim=2), dim=2)))
                gen_loss = gen_loss + var_loss
                total_var_loss += var_loss.detach() / acc
            gen_loss = gen_loss / acc
            gen_loss.backward()
            total_rec_loss += rec_loss.detach() / acc
            total_gen_loss += d_loss.detach() / acc
            total_hist_loss += histogram_loss.detach() / acc
        set_requires_grad(Disc, True)
        GAN._reduce_g()
        GAN.G_opt.step()

        # ---- one read-back per step
        stats = torch.stack([total_disc_loss, total_gen_loss, total_rec_loss, total_hist_loss, total_var_loss,
                             gp_val.reshape(()), q_val.reshape(())]).double()
        if ddp.is_dist():
            nan_flag = torch.isnan(stats[:2]).any().double().reshape(1)
            packed = torch.cat([stats, nan_flag])
            torch.distributed.all_reduce(packed[:7], op=torch.distributed.ReduceOp.SUM)
            torch.distributed.all_reduce(packed[7:], op=torch.distributed.ReduceOp.MAX)
            packed[:7] /= ddp.world_size()
            host = packed.cpu().numpy()
            has_nan = host[7] > 0 or np.isnan(host[:2]).any()
        else:
            host = stats.cpu().numpy()
            has_nan = bool(np.isnan(host[:2]).any())
        self.d_loss, self.g_loss, self.r_loss, self.h_loss = (float(v) for v in host[:4])
        if self.variance_loss is True:
            self.var_loss = float(host[4])
        if apply_gradient_penalty:
            self.last_gp_loss = float(host[5])
        self.q_loss = float(host[6])

        checkpoint_num = floor(self.steps / self.save_every)
        if has_nan:
            print(f'NaN detected for generator or discriminator. Loading from checkpoint #{checkpoint_num}')
            self.load(checkpoint_num)
            raise NanException
        if self.run_save and self.steps % self.save_every == 0:
            self.save(checkpoint_num)
        if self.run_evaluate and (self.steps % 1000 == 0 or (self.steps % 100 == 0 and self.steps < 2500)):
            self.evaluate(floor(self.steps / 1000))
        self.steps += 1
        self.av = None

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, num=0, image_batch=None, hist_batch=None, triple_hist=False, double_hist=False, resizing=None,
                 resizing_method=None, swapping_levels=1, pyramid_levels=5, level_blending=False, original_size=None,
                 input_image_name=None, original_image=None, post_recoloring=False, save_input=True):
        """Recolour a batch with its target histograms and write `<num>-generated.jpg` [+ `<num>-input.jpg`]: the
        reference's parameter list (ReHistoGAN/rehistoGAN.py:1076-1081).  The device part (encoder-decoder + head, the
        multi-histogram grids, the 'downscaling' resize of the written file) is implemented; the CPU / OpenCV / external
        post-processing options -- `resizing='upscaling'` (BGU.exe or the Laplacian-pyramid swap of
        utils/pyramid_upsampling.py) and `post_recoloring` (utils/color_transfer_MKL.py) -- are outside the hot path
         and raise NotImplementedError when asked for."""
        if resizing == 'upscaling' or post_recoloring:
            raise NotImplementedError("recoloringTrainer.evaluate: resizing='upscaling' (BGU / pyramid) and post_recoloring "
                                      'are CPU post-processing of the reference (utils/) outside the MI355X hot path')
        self.GAN.eval()
        if hist_batch is None or image_batch is None:
            batch = next(self.loader_evaluate)
            image_batch = batch['images'].to(self.device)
            hist_batch = batch['histograms'].to(self.device)
            img_bt_sz = image_batch.shape[0]
            extra = ['histograms2', 'histograms3'] if triple_hist is True else (['histograms2'] if double_hist is True else [])
            if extra:               # the same images against two / three target histograms (:1089-1100)
                image_batch = torch.cat([image_batch] * (len(extra) + 1), dim=0)
                hist_batch = torch.cat([hist_batch] + [batch[k].to(self.device) for k in extra], dim=0)
        else:
            img_bt_sz = image_batch.shape[0]
        noise = self.rng.image_noise(hist_batch.shape[0], image_batch.shape[-1])
        generated_images = self._recolor(image_batch, hist_batch, noise)
        if num is not None and self.is_main:
            from .data import save_image_grid
            ext = 'jpg' if not self.transparent else 'png'
            multi = double_hist is True or triple_hist is True
            num_rows = img_bt_sz if multi else int(np.ceil(np.sqrt(hist_batch.shape[0])))
            output_name = str(self.results_dir / self.name / f'{str(num)}-generated.{ext}')
            save_image_grid(generated_images, output_name, nrow=num_rows)
            if resizing == 'downscaling' and original_size is not None:
                from PIL import Image
                Image.open(output_name).resize((original_size[0], original_size[1])).save(output_name)
            if save_input is True:
                save_image_grid(image_batch[:img_bt_sz] if multi else image_batch,
                                str(self.results_dir / self.name / f'{str(num)}-input.{ext}'),
                                nrow=img_bt_sz if multi else num_rows)
        return generated_images

    def print_log(self):
        if not self.is_main:
            return
        msg = (f'\nG: {self.g_loss:.2f} | D: {self.d_loss:.2f} | GP: {self.last_gp_loss:.2f} | R: '
               f'{getattr(self, "r_loss", 0):.2f} | H: {getattr(self, "h_loss", 0):.2f}')
        if self.variance_loss is True:
            msg += f' | V: {self.var_loss:.2f}'
        print(msg + f' | Q: {self.q_loss:.2f}')

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
        if self.is_main:
            torch.save(self.GAN.state_dict(), self.model_name(num))
            self.write_config()

    def load(self, num=-1):
        self.load_config()
        name = num
        if num == -1:
            file_paths = [p for p in Path(self.models_dir / self.name).glob('model_*.pt')]
            saved_nums = sorted(map(lambda x: int(x.stem.split('_')[1]), file_paths))
            if len(saved_nums) == 0:
                return -1          # reference ReHistoGAN/rehistoGAN.py:1218: the CLI copies the pretrained HistoGAN head only then
            name = saved_nums[-1]
            print(f'continuing from previous epoch - {name}')
        self.steps = name * self.save_every
        self.GAN.load_state_dict(torch.load(self.model_name(name), map_location=self.device))
        weights_changed()
        return 0
