"""Drop-in for the reference's utils/diff_augment.py (`from utils.diff_augment import DiffAugment`,
histoGAN/histoGAN.py:36): same function names and parameter distributions; implementation: histogan_amd/augment.py over
the HIP kernels of include/hg_augment.h."""
import torch

from histogan_amd.augment import DiffAugment, _spatial_run, augment_color


def rand_brightness(x):
    return augment_color(x, torch.stack([torch.rand(x.size(0)) - 0.5, torch.ones(x.size(0)), torch.ones(x.size(0))], 1))


def rand_saturation(x):
    return augment_color(x, torch.stack([torch.zeros(x.size(0)), torch.rand(x.size(0)) * 2, torch.ones(x.size(0))], 1))


def rand_contrast(x):
    return augment_color(x, torch.stack([torch.zeros(x.size(0)), torch.ones(x.size(0)), torch.rand(x.size(0)) + 0.5], 1))


def rand_translation(x, ratio=0.125):
    return _spatial_run(x, ['translation'], ratios={'translation': dict(ratio=ratio)})


def rand_offset(x, ratio=1, ratio_h=1, ratio_v=1):
    return _spatial_run(x, ['offset'], ratios={'offset': dict(ratio=ratio, ratio_h=ratio_h, ratio_v=ratio_v)})


def rand_offset_h(x, ratio=1):
    return rand_offset(x, ratio=1, ratio_h=ratio, ratio_v=0)


def rand_offset_v(x, ratio=1):
    return rand_offset(x, ratio=1, ratio_h=0, ratio_v=ratio)


def rand_cutout(x, ratio=0.5):
    return _spatial_run(x, ['cutout'], ratios={'cutout': dict(ratio=ratio)})


AUGMENT_FNS = {
    'color': [rand_brightness, rand_saturation, rand_contrast],
    'offset': [rand_offset],
    'offset_h': [rand_offset_h],
    'offset_v': [rand_offset_v],
    'translation': [rand_translation],
    'cutout': [rand_cutout],
}

__all__ = ['DiffAugment', 'AUGMENT_FNS', 'rand_brightness', 'rand_saturation', 'rand_contrast', 'rand_translation',
           'rand_offset', 'rand_offset_h', 'rand_offset_v', 'rand_cutout']
