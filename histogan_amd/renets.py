"""ReHistoGAN networks on the MI355X kernels: same classes, constructor arguments,
forward signatures and state_dict key names as ReHistoGAN/rehistoGAN.py of the reference (:449-634), so its
checkpoints load.

Execution plan: every convolution (3x3, 1x1, 3x3 stride 2) is the fp32-MFMA implicit GEMM of hg_conv.h (bias and,
where one follows directly, LeakyReLU fused into its epilogue); InstanceNorm2d + LeakyReLU is one fused HIP op
(hg_recolor.h); the bilinear x2 upsamples are the polyphase kernel of hg_nets.h; the histogram-modulated
convolutions and the two-block recolouring head are the generator kernels of histogan_amd/nets.py.
"""
from math import log2

import torch
from torch import nn

from . import ops, reops
from .conv import conv2d_lrelu
from .nets import Conv2d, Conv2DMod, GeneratorBlock, HistVectorizer, _noise_t, leaky_relu


class RecoloringGAN(nn.Module):
    """The last two GeneratorBlocks of the HistoGAN generator, fed by the encoder-decoder (reference :449-482)."""

    def __init__(self, image_size, latent_dim, network_capacity=16, transparent=False):
        super().__init__()
        self.image_size = image_size
        self.latent_dim = latent_dim
        num_layers = int(log2(image_size) - 1)
        init_channels = 4 * network_capacity
        filters = [init_channels] + [network_capacity * (2 ** (i + 1)) for i in range(num_layers)][::-1]
        filters = filters[-3:]
        self.num_layers = 2
        self.blocks = nn.ModuleList([])
        for ind, (in_chan, out_chan) in enumerate(zip(filters[0:-1], filters[1:])):
            self.blocks.append(GeneratorBlock(latent_dim, in_chan, out_chan, upsample=True, upsample_rgb=ind != 1,
                                              rgba=transparent))

    def forward(self, x, rgb, hists, input_noise, latent1=None, latent2=None):
        rgb = None                                   # the decoder's rgb is discarded (reference :478)
        _noise_t(input_noise)
        x, rgb = self.blocks[0](x, rgb, hists, input_noise, latent=latent1)
        x, rgb = self.blocks[1](x, rgb, hists, input_noise, latent=latent2)
        return rgb


class EncoderBlock(nn.Module):
    def __init__(self, input_channels, filters):
        super().__init__()
        self.conv_res = Conv2d(input_channels, filters, 1)
        self.net = nn.Sequential(
            Conv2d(input_channels, filters, 3, padding=1), nn.InstanceNorm2d(filters), leaky_relu(),
            Conv2d(filters, filters, 3, padding=1), nn.InstanceNorm2d(filters), leaky_relu())
        self.downsample = Conv2d(filters, filters, 3, padding=1, stride=2)

    def forward(self, x):
        res = self.conv_res(x)
        # == self.net(x): Conv2d, InstanceNorm2d(affine=False, eps), LeakyReLU(0.2), twice
        x = reops.instnorm_lrelu(self.net[0](x), self.net[1].eps, 0.2)
        x = reops.instnorm_lrelu(self.net[3](x), self.net[4].eps, 0.2)
        x = x + res
        return self.downsample(x), x


def _conv_lrelu(seq, x):
    """Sequential(Conv2d, LeakyReLU(0.2)) as one launch."""
    return conv2d_lrelu(x, seq[0].weight, seq[0].bias, 0.2)


class DecoderBlock(nn.Module):
    def __init__(self, input_channels, filters, internal_hist=False, latent_dim=None):
        super().__init__()
        # kept for attribute parity; the HIP upsample kernel is used in forward
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)
        self.conv_res = Conv2d(input_channels, filters, 1)
        self.block1 = nn.Sequential(Conv2d(input_channels, input_channels, 3, padding=1), leaky_relu())
        self.block2 = nn.Sequential(Conv2d(input_channels * 2, filters, 3, padding=1), leaky_relu())
        self.conv_out_latent = nn.Sequential(Conv2d(filters, filters, 3, padding=1), leaky_relu())
        self.conv_out_rgb = Conv2d(filters, 3, 1)
        if internal_hist:
            self.to_latent = nn.Linear(latent_dim, input_channels)
            self.conv_latent = Conv2DMod(input_channels, input_channels, 3)
        else:
            self.to_latent = None
            self.conv_latent = None

    def forward(self, x, prev_rgb, prev_latent, h=None):
        curr_latent = _conv_lrelu(self.block1, x)
        if self.to_latent is not None:
            prev_latent = self.conv_latent(prev_latent, self.to_latent(h))
        processed_x = _conv_lrelu(self.block2, torch.cat((curr_latent, prev_latent), dim=1))
        x_res = self.conv_res(x)
        x = _conv_lrelu(self.conv_out_latent, x_res + processed_x)
        rgb = self.conv_out_rgb(x)
        if prev_rgb is not None:
            rgb = rgb + prev_rgb
        return ops.upsample2x(x), ops.upsample2x(rgb)


class RecoloringEncoderDecoder(nn.Module):
    def __init__(self, image_size, network_capacity=16, hist=64, latent_dim=512, style_depth=8,
                 skip_conn_to_GAN=False, internal_hist=False):
        super().__init__()
        self.image_size = image_size
        self.encoder_num_layers = int(log2(image_size) - 2)
        self.decoder_num_layers = int(log2(image_size) - 4)
        self.skip_conn_to_GAN = skip_conn_to_GAN
        self.internal_hist = internal_hist
        encoder_filters = [network_capacity] + [network_capacity * (2 ** (i + 1))
                                                for i in range(self.encoder_num_layers)]
        encoder_pairs = list(zip(encoder_filters[0:-1], encoder_filters[1:]))
        decoder_filters = encoder_filters[::-1]
        decoder_filters = decoder_filters[:-(self.encoder_num_layers - self.decoder_num_layers)]
        decoder_pairs = list(zip(decoder_filters[0:-1], decoder_filters[1:]))
        # the reference reverses `encoder_filters` in place before indexing it with [-3] / [-2] (:565-567, 575-590)
        rev = encoder_filters[::-1]

        self.encoder_blocks = nn.ModuleList([])
        self.decoder_blocks = nn.ModuleList([])
        self.decoder_mapping = Conv2d(decoder_filters[-1], 8 * network_capacity, 1)
        self.mapping = Conv2d(3, network_capacity, 3, padding=1)
        if self.skip_conn_to_GAN:
            if not self.internal_hist:
                self.hist_projection = HistVectorizer(hist, latent_dim, int(style_depth))
            self.to_latent_1 = nn.Linear(latent_dim, rev[-3])
            self.to_latent_2 = nn.Linear(latent_dim, rev[-2])
            self.conv_latent_1 = Conv2DMod(rev[-3], 2 ** 2 * network_capacity, 3)
            self.conv_latent_2 = Conv2DMod(rev[-2], 2 ** (2 - 1) * network_capacity, 3)
        for in_chan, out_chan in encoder_pairs:
            self.encoder_blocks.append(EncoderBlock(in_chan, out_chan))
        for in_chan, out_chan in decoder_pairs:
            self.decoder_blocks.append(DecoderBlock(in_chan, out_chan, internal_hist=self.internal_hist,
                                                    latent_dim=latent_dim))

    def forward(self, x, hists=None):
        if self.skip_conn_to_GAN:
            h_w_space = self.hist_projection(hists) if not self.internal_hist else hists
            h1 = self.to_latent_1(h_w_space)
            h2 = self.to_latent_2(h_w_space)
        x = self.mapping(x)
        x_list, x_list_up = [], []
        for block in self.encoder_blocks:
            x, xup = block(x)
            x_list.append(x)
            x_list_up.append(xup)
        x_list.reverse()
        if self.skip_conn_to_GAN:
            processed_latent_1 = self.conv_latent_1(x_list_up[1], h1)
            processed_latent_2 = self.conv_latent_2(x_list_up[0], h2)
        rgb = None
        for prev_latent, block in zip(x_list[:-2], self.decoder_blocks):
            x, rgb = block(x, rgb, prev_latent, h=hists)
        x = self.decoder_mapping(x)
        if self.skip_conn_to_GAN:
            return x, rgb, processed_latent_1, processed_latent_2
        return x, rgb
