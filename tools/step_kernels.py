#!/usr/bin/env python3
"""Run N HistoGAN train steps that all have the SAME step index (so all are gradient-penalty steps, or none is), for
kernel traces:   rocprofv3 --kernel-trace -d out -o t -- python tools/step_kernels.py --index 4 --steps 6"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--index', type=int, default=5)
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--batch', type=int, default=32)
a = ap.parse_args()
from histoGAN import Trainer  # noqa: E402

tmp = tempfile.mkdtemp()
tr = Trainer('p', tmp + '/r', tmp + '/m', 256, 16, batch_size=a.batch, hist_insz=150)
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src()
for _ in range(a.steps):
    tr.steps = a.index
    tr.train()
torch.cuda.synchronize()
print('done', a.index, a.steps)
