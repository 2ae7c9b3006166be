#!/usr/bin/env python3
"""Host-side enqueue time of one HistoGAN train step vs its wall time (is the step launch-bound?).

    python tools/host_probe.py [--batch 32] [--steps 12]
Marks: t0 = train() entered, tD = D phase enqueued (D grads all-reduce start), tG = G_opt.step enqueued (the last launch
before the read-back), t1 = train() returned (after the one device->host sync)."""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--steps', type=int, default=12)
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--cap', type=int, default=16)
ap.add_argument('--fused', action='store_true', help='GeneratorBlock.FUSED = True (one launch per generator stage)')
ap.add_argument('--profile-step', type=int, default=-1, help='cProfile this step index (host side)')
a = ap.parse_args()
from histoGAN import Trainer  # noqa: E402
from histogan_amd.nets import GeneratorBlock  # noqa: E402

GeneratorBlock.FUSED = a.fused

tmp = tempfile.mkdtemp()
tr = Trainer('p', tmp + '/r', tmp + '/m', a.size, a.cap, batch_size=a.batch, hist_insz=150)
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src()
for _ in range(5):
    tr.train()
marks = {}
gstep, dstart = tr.GAN.G_opt.step, tr.GAN._reduce_d.start


def g_step():
    gstep()
    marks['tG'] = time.perf_counter()


def d_start():
    dstart()
    marks['tD'] = time.perf_counter()


tr.GAN.G_opt.step, tr.GAN._reduce_d.start = g_step, d_start
rows = []
torch.cuda.synchronize()
prev_end = time.perf_counter()
for i in range(a.steps):
    t0 = time.perf_counter()
    if tr.steps == a.profile_step:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        tr.train()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)
    else:
        tr.train()
    t1 = time.perf_counter()
    rows.append((tr.steps - 1, (t0 - prev_end) * 1e3, (marks['tD'] - t0) * 1e3, (marks['tG'] - marks['tD']) * 1e3,
                 (t1 - marks['tG']) * 1e3, (t1 - t0) * 1e3))
    prev_end = t1
print('step | between calls | D-phase enqueue | G-phase enqueue | wait for GPU (read-back) | total   [ms]')
for r in rows:
    print('%4d | %8.3f | %8.2f | %8.2f | %8.2f | %8.2f' % r)
n = len(rows)
print('mean | %8.3f | %8.2f | %8.2f | %8.2f | %8.2f' % tuple(sum(r[k] for r in rows) / n for k in range(1, 6)))
