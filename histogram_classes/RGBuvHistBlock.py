"""Drop-in for the reference's histogram_classes/RGBuvHistBlock.py (same import path, ctor, forward).

`from histogram_classes.RGBuvHistBlock import RGBuvHistBlock` keeps working in the reference's
scripts (histoGAN.py:24, histoGAN/histoGAN.py:37, create_hist_*.py); forward runs the hand-written
gfx950 kernels of histogan_amd/csrc/hg_hist.hip through the C ABI of include/hg_hist.h.
"""
import torch
import torch.nn as nn

from histogan_amd.hist import HistConfig, run_block

EPS = 1e-6


class RGBuvHistBlock(nn.Module):
  def __init__(self, h=64, insz=150, resizing='interpolation',
               method='inverse-quadratic', sigma=0.02, intensity_scale=True,
               hist_boundary=None, green_only=False, device='cuda'):
    """Computes the RGB-uv histogram feature of a given image batch.

    Same arguments as the reference class (RGBuvHistBlock.py:29-55): h bins per axis; images
    larger than insz are resized ('interpolation' = bilinear to insz x insz, 'sampling' = h x h
    strided samples); method in {'thresholding', 'RBF', 'inverse-quadratic'}; sigma of the
    RBF / inverse-quadratic kernel; intensity_scale (I_y weighting); hist_boundary (default
    [-3, 3], sorted in place like the reference); green_only (only the log(g/r), log(g/b) plane).
    `device`: a GPU runs the HIP kernels; 'cpu' (the reference Dataset's use inside DataLoader workers) runs
    histogan_amd/hist_cpu.py -- PyTorch CPU ops, no HIP call (histogan_amd.hist.run_block).
    """
    super(RGBuvHistBlock, self).__init__()
    self.h = h
    self.insz = insz
    self.device = device
    self.resizing = resizing
    self.method = method
    self.intensity_scale = intensity_scale
    self.green_only = green_only
    if hist_boundary is None:
      hist_boundary = [-3, 3]
    hist_boundary.sort()
    self.hist_boundary = hist_boundary
    if self.method == 'thresholding':
      self.eps = (abs(hist_boundary[0]) + abs(hist_boundary[1])) / h
    else:
      self.sigma = sigma

  def _config(self):
    return HistConfig(h=self.h, insz=self.insz, resizing=self.resizing, method=self.method,
                      sigma=getattr(self, 'sigma', 0.02), intensity_scale=self.intensity_scale,
                      hist_boundary=list(self.hist_boundary), green_only=self.green_only)

  def forward(self, x, pre_relu=False):
    """x: float (B, C>=3, H, W) -> float32 (B, 3 or 1, h, h), L1-normalised per image, on `device`.
    pre_relu=True: the value and gradient of forward(F.relu(x)) -- the train step's call (histoGAN/histoGAN.py:955) --
    with the relu folded into the kernel's clamp mask (an extension; the reference signature is forward(x))."""
    return run_block(x, self._config(), self.device, 'RGBuvHistBlock', pre_relu)
