"""Inputs of the BASELINE-shape golden cases (tests/golden/big_*.npz): regenerated from their seeds.

Shared by tests/golden/make_golden_big.py (which runs the unmodified reference on them in the authoring container)
and by the parity tests (which run on the GPU box, where only the committed reference OUTPUTS exist)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def make_input(spec):
    kind = spec['kind']
    g = torch.Generator().manual_seed(spec['seed'])
    if kind == 'uniform':
        return torch.rand(*spec['shape'], generator=g)
    if kind == 'genlike':                                   # raw generator-like output; callers apply F.relu
        return 0.5 + 0.5 * torch.randn(*spec['shape'], generator=g)
    raise ValueError(kind)


def u8_to_tensor(u8):
    """torchvision ToTensor restated: HWC uint8 -> 1xCxHxW float32 in [0, 1]."""
    return torch.from_numpy(np.ascontiguousarray((u8.astype(np.float32) / 255.0).transpose(2, 0, 1))[None])


def big_names():
    with open(os.path.join(GOLDEN_DIR, 'BIG_INDEX.json')) as f:
        return json.load(f)


def load_big(name):
    """-> dict with 'spec' (dict), 'x' (raw input tensor, before the optional relu), the stored reference outputs."""
    z = np.load(os.path.join(GOLDEN_DIR, f'big_{name}.npz'))
    rec = {k: z[k] for k in z.files}
    rec['spec'] = spec = json.loads(str(rec['spec']))
    rec['x'] = u8_to_tensor(rec['x_u8']) if 'x_u8' in rec else make_input(spec['x'])
    return rec


def rand_grad_out(shape):
    g = torch.Generator().manual_seed(99)
    return torch.rand(shape, generator=g) - 0.3
