#!/usr/bin/env python3
"""Golden vectors for DiffAugment from the UNMODIFIED utils/diff_augment.py of the
reference: its torch.rand / torch.randint / random.randint draws are recorded while it runs, so the same parameters
can be replayed through oracle/diff_augment.py and the HIP kernels.

    python tests/golden/make_golden_augment.py      # writes tests/golden/augment.npz
"""
import os
import random
import sys

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.diff_augment import cutout_box  # noqa: E402

ID = [0, 0, 0, 0, 0, 1, 0, 1, 0]


def main():
    sys.path.insert(0, REF)
    import utils.diff_augment as R
    sys.path.pop(0)
    log = []
    real_rand, real_randint, real_pyrandint = torch.rand, torch.randint, random.randint

    def rec(kind, fn):
        def f(*a, **k):
            v = fn(*a, **k)
            log.append((kind, v.clone().reshape(-1).tolist() if torch.is_tensor(v) else v))
            return v
        return f

    torch.rand, torch.randint = rec('rand', real_rand), rec('randint', real_randint)
    R.random.randint = rec('pyrandint', real_pyrandint)
    torch.manual_seed(0)
    random.seed(0)
    out = {}
    cases = {'translation': (3, 3, 20, 28), 'cutout': (4, 3, 16, 16), 'cutout_odd': (3, 3, 15, 21),
             'offset': (3, 3, 12, 12), 'offset_h': (2, 3, 12, 10), 'offset_v': (2, 3, 10, 12),
             'color': (3, 3, 9, 11), 'translation+cutout': (4, 3, 32, 32), 'color+translation+cutout': (2, 3, 16, 16),
             'cutout+translation': (3, 3, 16, 16)}
    for name, shape in cases.items():
        types = name.replace('_odd', '').split('+')
        B, C, H, W = shape
        x = real_rand(shape).requires_grad_(True)
        del log[:]
        y = R.DiffAugment(x, types=types)
        go = torch.randn_like(y)
        gx, = torch.autograd.grad(y, x, go)
        draws = list(log)
        # replay table: one entry per augmentation in `types`: ('spatial', rows) or ('color', rows)
        steps, it = [], iter(draws)
        for t in types:
            rows = [list(ID) for _ in range(B)]
            if t == 'color':
                br, sa, co = next(it)[1], next(it)[1], next(it)[1]
                steps.append(('color', [[br[b] - 0.5, sa[b] * 2, co[b] + 0.5] for b in range(B)]))
                continue
            if t == 'translation':
                tx, ty = next(it)[1], next(it)[1]
                for b in range(B):
                    rows[b][3], rows[b][4] = int(tx[b]), int(ty[b])
            elif t == 'cutout':
                ox, oy = next(it)[1], next(it)[1]
                for b in range(B):
                    rows[b][5:9] = cutout_box(int(ox[b]), int(oy[b]), H, W)
            else:   # offset*: per image value_h = randint*2 - max_h (rolls W), value_v (rolls H)
                rh, rv = (0 if t == 'offset_v' else 1), (0 if t == 'offset_h' else 1)
                max_h, max_v = int(H * rh), int(W * rv)
                for b in range(B):
                    vh, vv = next(it)[1] * 2 - max_h, next(it)[1] * 2 - max_v
                    rows[b][2], rows[b][1] = vh, vv
            steps.append(('spatial', rows))
        assert next(it, None) is None, name
        out[f'{name}/x'] = x.detach().numpy(); out[f'{name}/y'] = y.detach().numpy()
        out[f'{name}/go'] = go.numpy(); out[f'{name}/gx'] = gx.numpy()
        for i, (kind, rows) in enumerate(steps):
            out[f'{name}/step{i}_{kind}'] = np.array(rows, dtype=np.float64 if kind == 'color' else np.int32)
    torch.rand, torch.randint, R.random.randint = real_rand, real_randint, real_pyrandint
    # AugWrapper's horizontal flip (histoGAN/histoGAN.py:312-315)
    x = torch.rand(2, 3, 6, 7)
    out['flip/x'] = x.numpy(); out['flip/y'] = torch.flip(x, dims=(3,)).numpy()
    np.savez_compressed(os.path.join(HERE, 'augment.npz'), **out)
    print('wrote augment.npz with', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'augment.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
