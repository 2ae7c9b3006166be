import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_names():
    with open(os.path.join(GOLDEN_DIR, 'INDEX.json')) as f:
        return json.load(f)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f'hist_{name}.npz'))
    rec = {k: z[k] for k in z.files}
    rec['kwargs'] = json.loads(str(rec['kwargs']))
    return rec


def relmax(a, b):
    """max|a-b| / max|b| -- the parity norm of SURVEY.md section 8c."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
