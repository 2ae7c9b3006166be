#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference in the authoring container.

    python tests/golden/make_golden.py            # writes tests/golden/hist_*.npz

The reference (/root/reference) exists only in the authoring container, never on
the GPU box, so the vectors are committed.  Each .npz holds the input image(s),
the ctor kwargs (json), the reference forward output, the reference autograd
gradient for a seeded upstream gradient, and -- for the Hellinger cases -- the
reference loss value (formula of histoGAN/histoGAN.py:957-960 with alpha=1, i.e.
Histogram_loss.ipynb:415-417) and its gradient.

Nothing in tests/ imports the reference at test time.
"""
import json
import os
import sys

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.path.insert(0, REF)
    from histogram_classes.RGBuvHistBlock import RGBuvHistBlock  # noqa: E402
    sys.path.pop(0)
    return RGBuvHistBlock


def jpeg_crop(path, y0, x0, hh, ww, step=1):
    from PIL import Image
    im = np.asarray(Image.open(path).convert('RGB'), dtype=np.float32) / 255.0  # ToTensor
    im = im[y0:y0 + hh * step:step, x0:x0 + ww * step:step]
    return np.ascontiguousarray(im.transpose(2, 0, 1))[None]


def cases():
    g = torch.Generator().manual_seed(1234)

    def rand(*s):
        return torch.rand(*s, generator=g)

    def randn_img(*s):  # generator-like: exact zeros after relu, values > 1
        return torch.relu(0.5 + 0.5 * torch.randn(*s, generator=g))

    c = []
    c.append(('iq_h64_b2_48', rand(2, 3, 48, 48), dict(h=64), True))
    c.append(('iq_h16_b3_20x28', rand(3, 3, 20, 28), dict(h=16), True))
    c.append(('iq_h32_genlike', randn_img(2, 3, 40, 40), dict(h=32), True))
    c.append(('iq_h64_interp_64to40', rand(2, 3, 64, 64), dict(h=64, insz=40), True))
    c.append(('iq_h16_interp_nonsq_37x53to24', rand(1, 3, 37, 53), dict(h=16, insz=24), True))
    c.append(('iq_h16_interp_mixed_20x50to32', rand(1, 3, 20, 50), dict(h=16, insz=32), True))
    c.append(('iq_h16_sampling_70x45', rand(2, 3, 70, 45), dict(h=16, insz=32, resizing='sampling'), True))
    c.append(('rbf_h32_b2_40', rand(2, 3, 40, 40), dict(h=32, method='RBF'), False))
    c.append(('rbf_h16_sigma_0p1', rand(1, 3, 32, 32), dict(h=16, method='RBF', sigma=0.1), False))
    c.append(('thr_h64_b2_48', rand(2, 3, 48, 48), dict(h=64, method='thresholding'), False))
    c.append(('thr_h16_genlike', randn_img(2, 3, 32, 32), dict(h=16, method='thresholding'), False))
    c.append(('iq_h32_green_only', rand(2, 3, 32, 32), dict(h=32, green_only=True), True))
    c.append(('iq_h32_no_intensity', rand(2, 3, 32, 32), dict(h=32, intensity_scale=False), True))
    c.append(('iq_h16_c4', rand(2, 4, 24, 24), dict(h=16), True))
    c.append(('iq_h16_asym_boundary', rand(2, 3, 24, 24), dict(h=16, hist_boundary=[-2, 4]), True))
    c.append(('iq_h24_odd_h', rand(1, 3, 24, 24), dict(h=24), True))
    c.append(('iq_h128_b1_32', rand(1, 3, 32, 32), dict(h=128), True))
    c.append(('iq_h64_sigma_0p05', rand(1, 3, 32, 32), dict(h=64, sigma=0.05), True))
    # constant / saturated colours (u = +-13.8), tiny image
    edge = torch.zeros(5, 3, 2, 2)
    edge[1] = 1.0
    edge[2, 0] = 1.0                      # pure red
    edge[3] = torch.tensor([0.25, 0.5, 0.75]).view(3, 1, 1)
    edge[4] = rand(3, 2, 2) * 3 - 1      # out-of-range values (clamp mask)
    c.append(('iq_h16_edge_colours_2x2', edge, dict(h=16), False))
    # real photographs shipped with the reference (ToTensor restated as /255 -> CHW)
    c.append(('iq_h64_jpeg_target2_crop', torch.from_numpy(
        jpeg_crop(os.path.join(REF, 'target_images', '2.jpg'), 40, 60, 64, 64, step=2)), dict(h=64), True))
    c.append(('iq_h64_jpeg_target5_interp', torch.from_numpy(
        jpeg_crop(os.path.join(REF, 'target_images', '5.jpg'), 0, 0, 96, 80, step=2)), dict(h=64, insz=48), True))
    return c


def main():
    RGBuvHistBlock = load_reference()
    torch.manual_seed(0)
    index = []
    for name, x, kw, hell in cases():
        x = x.float().contiguous()
        kw_ctor = {k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()}
        blk = RGBuvHistBlock(device='cpu', **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        xr = x.clone().requires_grad_(True)
        out = blk(xr)
        g = torch.Generator().manual_seed(99)
        grad_out = torch.rand(out.shape, generator=g) - 0.3
        (gx,) = torch.autograd.grad(out, xr, grad_out, retain_graph=hell)
        rec = dict(x=x.numpy(), hist=out.detach().numpy(), grad_out=grad_out.numpy(),
                   grad_x=gx.numpy(), kwargs=json.dumps(kw_ctor))
        if hell:
            # target histogram: the reference on a second image of the same shape
            gt = torch.Generator().manual_seed(7)
            xt = torch.rand(x.shape, generator=gt)
            with torch.no_grad():
                tgt = blk(xt)
            loss = (1 / np.sqrt(2.0)) * (torch.sqrt(torch.sum(
                torch.pow(torch.sqrt(tgt) - torch.sqrt(out), 2)))) / out.shape[0]
            (gxl,) = torch.autograd.grad(loss, xr)
            rec.update(target_hist=tgt.numpy(), hell_loss=np.float64(loss.item()),
                       hell_grad_x=gxl.numpy())
        np.savez_compressed(os.path.join(HERE, f'hist_{name}.npz'), **rec)
        index.append(name)
        print(f'{name:40s} x{tuple(x.shape)} -> hist{tuple(out.shape)} sum={out.sum().item():.6f}')
    with open(os.path.join(HERE, 'INDEX.json'), 'w') as f:
        json.dump(index, f, indent=1)


if __name__ == '__main__':
    main()
