#!/usr/bin/env python3
"""Golden vectors of the reference's one-plane histogram blocks (rgChromaHistBlock, LabHistBlock), produced by
running the UNMODIFIED reference classes in the authoring container:

    python tests/golden/make_golden_planes.py        # writes tests/golden/plane_*.npz + PLANES_INDEX.json

Each .npz: input, ctor kwargs (json), block name, reference forward output, reference autograd gradient for a
seeded upstream gradient.  Nothing in tests/ imports the reference at test time.
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
    from histogram_classes.rgChromaHistBlock import rgChromaHistBlock  # noqa: E402
    from histogram_classes.LabHistBlock import LabHistBlock  # noqa: E402
    sys.path.pop(0)
    return {'rgchroma': rgChromaHistBlock, 'direct': LabHistBlock}


def cases():
    g = torch.Generator().manual_seed(4321)
    rand = lambda *s: torch.rand(*s, generator=g)
    gen = lambda *s: torch.relu(0.5 + 0.5 * torch.randn(*s, generator=g))     # exact zeros and values > 1
    c = []
    for proj in ('rgchroma', 'direct'):
        c.append((f'{proj}_iq_h64_b2_40', proj, rand(2, 3, 40, 40), dict(h=64)))
        c.append((f'{proj}_iq_h16_intensity_genlike', proj, gen(2, 3, 24, 28), dict(h=16, intensity_scale=True)))
        c.append((f'{proj}_iq_h32_interp_50to32', proj, rand(2, 3, 50, 44), dict(h=32, insz=32)))
        c.append((f'{proj}_iq_h16_sampling', proj, rand(1, 3, 40, 36), dict(h=16, insz=24, resizing='sampling')))
        c.append((f'{proj}_rbf_h32_sigma0p05', proj, rand(1, 3, 32, 32), dict(h=32, method='RBF', sigma=0.05)))
        c.append((f'{proj}_thr_h16', proj, rand(2, 3, 24, 24), dict(h=16, method='thresholding', intensity_scale=True)))
        c.append((f'{proj}_iq_h16_boundary_m1_1', proj, rand(1, 4, 20, 20), dict(h=16, hist_boundary=[-1, 1], intensity_scale=True)))
        c.append((f'{proj}_iq_h128', proj, rand(1, 3, 24, 24), dict(h=128, intensity_scale=True)))
    return c


def main():
    ref = load_reference()
    index = []
    for name, proj, x, kw in cases():
        x = x.float().contiguous()
        blk = ref[proj](device='cpu', **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        xr = x.clone().requires_grad_(True)
        out = blk(xr)
        go = torch.rand(out.shape, generator=torch.Generator().manual_seed(99)) - 0.3
        (gx,) = torch.autograd.grad(out, xr, go)
        np.savez_compressed(os.path.join(HERE, f'plane_{name}.npz'), x=x.numpy(), hist=out.detach().numpy(),
                            grad_out=go.numpy(), grad_x=gx.numpy(), kwargs=json.dumps(kw), projection=proj)
        index.append(name)
        print(f'{name:44s} x{tuple(x.shape)} -> {tuple(out.shape)} sum={out.sum().item():.6f}')
    with open(os.path.join(HERE, 'PLANES_INDEX.json'), 'w') as f:
        json.dump(index, f, indent=1)


if __name__ == '__main__':
    main()
