#!/usr/bin/env python3
"""Golden vectors at the shapes BASELINE.json's numbers are quoted on, from the UNMODIFIED reference.

    python tests/golden/make_golden_big.py        # writes tests/golden/big_*.npz + BIG_INDEX.json

Round-1 fixtures stop at 64x64 pixels per image; the bench line runs 256x256 (65 536 pixels per image, 1 024-pixel
per-wave chunks, 16 split-K slabs).  These cases put the reference itself (`/root/reference/histogram_classes/
RGBuvHistBlock.py:75-228`, imported as is, device='cpu') on exactly those inputs:

  c1_4x128            SURVEY 8d C1 / BASELINE configs[0]: manual_seed(0) rand(4,3,128,128), target manual_seed(1),
                      h=64, insz=150, inverse-quadratic, sigma 0.02, Hellinger loss of Histogram_loss.ipynb:415-417
  c2_2x256_uniform    configs[1] per-image shape: rand(2,3,256,256), insz=256 (ipynb:394) -> N = 65 536
  c2_2x256_genlike    same, x = 0.5+0.5*randn fed through F.relu first (histoGAN/histoGAN.py:955): exact zeros, values > 1
  trainer_2x256to150  the trainer default: 256^2 -> 150^2 bilinear, generator-like input
  h128_2x256to150     configs[4]'s histogram: h=128
  jpeg1024to150       target_images/1.jpg (1024^2) -> 150^2, the Dataset's use (histoGAN/histoGAN.py:296-302)
  rbf_1x256 / thr_1x256   the other two kernels at N = 65 536

Inputs are regenerated from their seed at test time (same torch build on the GPU box); the photograph is stored as
decoded uint8.  Stored: reference histogram, Hellinger loss + its input gradient (upstream gradient of the train
step), and for some cases the gradient for a seeded random upstream gradient.  The 1024^2 gradient is stored on a
stride-3 pixel lattice plus per-channel sums (12 MB otherwise).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.path.insert(0, REF)
    from histogram_classes.RGBuvHistBlock import RGBuvHistBlock  # noqa: E402
    sys.path.pop(0)
    return RGBuvHistBlock


sys.path.insert(0, os.path.dirname(HERE))
from bigcases import make_input, rand_grad_out, u8_to_tensor  # noqa: E402  (tests/bigcases.py)


CASES = [
    dict(name='c1_4x128', x=dict(kind='uniform', seed=0, shape=[4, 3, 128, 128]),
         t=dict(kind='uniform', seed=1, shape=[4, 3, 128, 128]), kw=dict(h=64, insz=150), rand_grad=True),
    dict(name='c2_2x256_uniform', x=dict(kind='uniform', seed=0, shape=[2, 3, 256, 256]),
         t=dict(kind='uniform', seed=1, shape=[2, 3, 256, 256]), kw=dict(h=64, insz=256)),
    dict(name='c2_2x256_genlike', x=dict(kind='genlike', seed=2, shape=[2, 3, 256, 256]), relu=True,
         t=dict(kind='uniform', seed=1, shape=[2, 3, 256, 256]), kw=dict(h=64, insz=256)),
    dict(name='trainer_2x256to150', x=dict(kind='genlike', seed=3, shape=[2, 3, 256, 256]), relu=True,
         t=dict(kind='uniform', seed=1, shape=[2, 3, 256, 256]), kw=dict(h=64, insz=150)),
    dict(name='h128_2x256to150', x=dict(kind='uniform', seed=4, shape=[2, 3, 256, 256]),
         t=dict(kind='uniform', seed=1, shape=[2, 3, 256, 256]), kw=dict(h=128, insz=150)),
    dict(name='jpeg1024to150', x=dict(kind='jpeg', file='target_images/1.jpg'),
         t=dict(kind='jpeg', file='target_images/4.jpg'), kw=dict(h=64, insz=150), grad_stride=3),
    dict(name='rbf_1x256', x=dict(kind='uniform', seed=5, shape=[1, 3, 256, 256]), t=None,
         kw=dict(h=64, insz=256, method='RBF'), rand_grad=True),
    dict(name='thr_1x256', x=dict(kind='uniform', seed=6, shape=[1, 3, 256, 256]), t=None,
         kw=dict(h=64, insz=256, method='thresholding'), rand_grad=True),
]


def load_jpeg_u8(rel):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(REF, rel)).convert('RGB'), dtype=np.uint8)


def main():
    RGBuvHistBlock = load_reference()
    index = []
    for c in CASES:
        blk = RGBuvHistBlock(device='cpu', **c['kw'])
        rec = dict(spec=json.dumps({k: v for k, v in c.items()}))
        if c['x']['kind'] == 'jpeg':
            u8 = load_jpeg_u8(c['x']['file'])
            rec['x_u8'] = u8
            x = u8_to_tensor(u8)
            tu8 = load_jpeg_u8(c['t']['file'])
            xt = u8_to_tensor(tu8)
        else:
            x = make_input(c['x'])
            xt = make_input(c['t']) if c['t'] else None
        xr = x.clone().requires_grad_(True)
        out = blk(F.relu(xr) if c.get('relu') else xr)
        rec['hist'] = out.detach().numpy()
        if c.get('rand_grad'):
            grad_out = rand_grad_out(out.shape)
            (gx,) = torch.autograd.grad(out, xr, grad_out, retain_graph=True)
            rec['grad_x'] = gx.numpy()
        if xt is not None:
            with torch.no_grad():
                tgt = blk(xt)
            loss = (1 / np.sqrt(2.0)) * (torch.sqrt(torch.sum(
                torch.pow(torch.sqrt(tgt) - torch.sqrt(out), 2)))) / out.shape[0]
            (gxl,) = torch.autograd.grad(loss, xr)
            gxl = gxl.numpy()
            rec['target_hist'] = tgt.numpy()
            rec['hell_loss'] = np.float64(loss.item())
            s = c.get('grad_stride')
            if s:
                rec['hell_grad_x_lattice'] = np.ascontiguousarray(gxl[:, :, ::s, ::s])
                rec['hell_grad_x_chansum'] = gxl.astype(np.float64).sum(axis=(2, 3))
                rec['hell_grad_x_absmax'] = np.float64(np.abs(gxl).max())
            else:
                rec['hell_grad_x'] = gxl
        np.savez_compressed(os.path.join(HERE, f'big_{c["name"]}.npz'), **rec)
        index.append(c['name'])
        print(f'{c["name"]:24s} x{tuple(x.shape)} -> hist{tuple(out.shape)} sum={out.sum().item():.6f}'
              + (f' loss={rec["hell_loss"]:.8f}' if 'hell_loss' in rec else ''), flush=True)
    with open(os.path.join(HERE, 'BIG_INDEX.json'), 'w') as f:
        json.dump(index, f, indent=1)


if __name__ == '__main__':
    main()
