"""Drop-in for the reference's histogram_classes/LabHistBlock.py (same import path, ctor, forward): the Lab (a, b)
histogram of an image batch, on the gfx950 kernels of histogan_amd/csrc/hg_hist.hip (projection 'direct' of
include/hg_hist.h: one plane, shared clamp / resize / soft-binning / normalisation code with the RGB-uv block).
"""
import torch
import torch.nn as nn

from histogan_amd.hist import HistConfig, run_block

EPS = 1e-6


class LabHistBlock(nn.Module):
  def __init__(self, h=64, insz=150, resizing='interpolation',
               method='inverse-quadratic', sigma=0.02, intensity_scale=False,
               hist_boundary=None, device='cuda'):
    """Same arguments as the reference class (LabHistBlock.py:30-71): h bins per axis; images larger than insz
    are resized ('interpolation' / 'sampling'); method in {'thresholding', 'RBF', 'inverse-quadratic'}; sigma;
    intensity_scale (weight = the L channel); hist_boundary (default [0, 1], sorted in place).  `device` must be a GPU."""
    super(LabHistBlock, self).__init__()
    self.h = h
    self.insz = insz
    self.device = device
    self.resizing = resizing
    self.method = method
    self.intensity_scale = intensity_scale
    if hist_boundary is None:
      hist_boundary = [0, 1]
    hist_boundary.sort()
    self.hist_boundary = hist_boundary
    if self.method == 'thresholding':
      self.eps = (abs(hist_boundary[0]) + abs(hist_boundary[1])) / h
    else:
      self.sigma = sigma

  def forward(self, x):
    """x: float (B, C>=3, H, W) -> float32 (B, 1, h, h), L1-normalised per image, on `device`."""
    cfg = HistConfig(h=self.h, insz=self.insz, resizing=self.resizing, method=self.method,
                     sigma=getattr(self, 'sigma', 0.02), intensity_scale=self.intensity_scale,
                     hist_boundary=list(self.hist_boundary), projection='direct')
    return run_block(x, cfg, self.device, 'LabHistBlock')
