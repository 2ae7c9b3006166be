"""Functional torch restatement of the ReHistoGAN networks and losses.
TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

State dicts use the reference's parameter names (`encoder_blocks.0.net.0.weight`, ...), so golden weights
from the reference classes plug in directly.  Pinned against goldens produced by the unmodified reference
classes (tests/golden/make_golden_rehistogan.py -> rehistogan_small.npz; tests/test_oracle_rehistogan_golden.py).

Follows (cites relative to the reference root, file ReHistoGAN/rehistoGAN.py):
* encoder_block ............. :485-504  (conv, InstanceNorm2d(affine=False, eps=1e-5), LeakyReLU(0.2) x2, + 1x1 residual,
                                         3x3 stride-2 downsample; returns (downsampled, full-resolution))
* decoder_block ............. :507-546
* encoder_decoder ........... :549-634  (returns x, rgb[, latent_1, latent_2])
* recoloring_head ........... :449-482  (RecoloringGAN: the last two GeneratorBlocks of the HistoGAN generator)
* gaussian_kernel ........... :207-225
* rec_loss .................. :279-326  ('L1' | '1st gradient' | '2nd gradient')
* variance_loss ............. :1022-1029
* histogram_loss ............ :1013-1016
"""
from math import log2, pi

import torch
import torch.nn.functional as F

from . import histogan_nets as N

lrelu = N.lrelu


def _conv(sd, name, x, padding=0, stride=1):
    return F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], padding=padding, stride=stride)


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def encoder_block(sd, pre, x):
    res = _conv(sd, pre + 'conv_res', x)
    y = lrelu(_inorm(_conv(sd, pre + 'net.0', x, padding=1)))
    y = lrelu(_inorm(_conv(sd, pre + 'net.3', y, padding=1)))
    y = y + res
    return _conv(sd, pre + 'downsample', y, padding=1, stride=2), y


def decoder_block(sd, pre, x, prev_rgb, prev_latent, h=None):
    curr = lrelu(_conv(sd, pre + 'block1.0', x, padding=1))
    if pre + 'to_latent.weight' in sd:
        prev_latent = N.conv2d_mod(prev_latent, N._lin(sd, pre + 'to_latent', h), sd[pre + 'conv_latent.weight'])
    processed = lrelu(_conv(sd, pre + 'block2.0', torch.cat((curr, prev_latent), dim=1), padding=1))
    x = lrelu(_conv(sd, pre + 'conv_out_latent.0', _conv(sd, pre + 'conv_res', x) + processed, padding=1))
    rgb = _conv(sd, pre + 'conv_out_rgb', x)
    if prev_rgb is not None:
        rgb = rgb + prev_rgb
    return N._up2(x), N._up2(rgb)


def encoder_decoder(sd, x, hists, image_size, skip_conn_to_GAN=False, internal_hist=False):
    n_enc, n_dec = int(log2(image_size) - 2), int(log2(image_size) - 4)
    if skip_conn_to_GAN:
        if not internal_hist:
            hp = {k[len('hist_projection.'):]: v for k, v in sd.items() if k.startswith('hist_projection.')}
            hw = N.vectorizer(hp, hists, 'fcs')
        else:
            hw = hists
        h1, h2 = N._lin(sd, 'to_latent_1', hw), N._lin(sd, 'to_latent_2', hw)
    x = _conv(sd, 'mapping', x, padding=1)
    downs, ups = [], []
    for i in range(n_enc):
        x, xup = encoder_block(sd, f'encoder_blocks.{i}.', x)
        downs.append(x)
        ups.append(xup)
    downs.reverse()
    if skip_conn_to_GAN:
        lat1 = N.conv2d_mod(ups[1], h1, sd['conv_latent_1.weight'])
        lat2 = N.conv2d_mod(ups[0], h2, sd['conv_latent_2.weight'])
    rgb = None
    for i, prev_latent in zip(range(n_dec), downs[:-2]):
        x, rgb = decoder_block(sd, f'decoder_blocks.{i}.', x, rgb, prev_latent, h=hists)
    x = _conv(sd, 'decoder_mapping', x)
    return (x, rgb, lat1, lat2) if skip_conn_to_GAN else (x, rgb)


def _generator_block_latent(sd, pre, x, prev_rgb, istyle, inoise, upsample_rgb, latent):
    x = N._up2(x)
    inoise = inoise[:, :x.shape[2], :x.shape[3], :]
    n1 = N._lin(sd, pre + 'to_noise1', inoise).permute(0, 3, 2, 1)
    n2 = N._lin(sd, pre + 'to_noise2', inoise).permute(0, 3, 2, 1)
    x = lrelu(N.conv2d_mod(x, N._lin(sd, pre + 'to_style1', istyle), sd[pre + 'conv1.weight']) + n1)
    if latent is not None:
        x = x + latent
    x = lrelu(N.conv2d_mod(x, N._lin(sd, pre + 'to_style2', istyle), sd[pre + 'conv2.weight']) + n2)
    return x, N.rgb_block(sd, pre + 'to_rgb.', x, prev_rgb, istyle, upsample_rgb)


def recoloring_head(sd, x, hists, noise, latent1=None, latent2=None):
    """RecoloringGAN.forward(x, rgb, hists, input_noise, latent1, latent2): the incoming rgb is DISCARDED (:478)."""
    x, rgb = _generator_block_latent(sd, 'blocks.0.', x, None, hists, noise, True, latent1)
    x, rgb = _generator_block_latent(sd, 'blocks.1.', x, rgb, hists, noise, False, latent2)
    return rgb


def gaussian_kernel(kernel_size=15, sigma=3, channels=3):
    xc = torch.arange(kernel_size)
    xg = xc.repeat(kernel_size).view(kernel_size, kernel_size)
    xy = torch.stack([xg, xg.t()], dim=-1).float()
    mean, var = (kernel_size - 1) / 2., sigma ** 2.
    k = (1. / (2. * pi * var)) * torch.exp(-torch.sum((xy - mean) ** 2., dim=-1) / (2 * var))
    k = k / torch.sum(k)
    return k.view(1, 1, kernel_size, kernel_size).repeat(channels, 1, 1, 1)


def gaussian_op(x, kernel):
    return F.conv2d(x, kernel, groups=x.shape[1])       # no padding: (H-14, W-14)


SOBEL_X = [[1, 0, -1], [2, 0, -2], [1, 0, -1]]
SOBEL_Y = [[1, 2, 1], [0, 0, 0], [-1, -2, -1]]
LAPLACIAN = [[0, 1, 0], [1, -4, 1], [0, 1, 0]]


def _stencil(x, taps):
    k = torch.tensor(taps, dtype=torch.float32).unsqueeze(0).expand(1, 3, 3, 3).to(x)
    return F.conv2d(x, k, stride=1, padding=1)


def rec_loss(kind, inp, target):
    if kind == 'L1':
        return torch.mean(torch.abs(inp - target))
    if kind == '1st gradient':
        gi = torch.sqrt(_stencil(inp, SOBEL_X) ** 2 + _stencil(inp, SOBEL_Y) ** 2)
        gt = torch.sqrt(_stencil(target, SOBEL_X) ** 2 + _stencil(target, SOBEL_Y) ** 2)
        return torch.mean(torch.abs(gi - gt))
    if kind == '2nd gradient':
        return torch.mean(torch.abs(_stencil(inp, LAPLACIAN) - _stencil(target, LAPLACIAN)))
    raise ValueError(kind)


def variance_loss(beta, hist_batch, input_histograms, image_batch, generated, kernel):
    ig, gg = gaussian_op(image_batch, kernel), gaussian_op(generated, kernel)
    return -1 * (beta / 10) * torch.sum(torch.abs(hist_batch - input_histograms)) * torch.mean(
        torch.abs(torch.std(torch.std(ig, dim=2), dim=2) - torch.std(torch.std(gg, dim=2), dim=2)))


def histogram_loss(alpha, hist_batch, generated_histograms):
    return alpha * (1 / 2 ** 0.5) * torch.sqrt(torch.sum(
        (torch.sqrt(hist_batch) - torch.sqrt(generated_histograms)) ** 2)) / hist_batch.shape[0]
