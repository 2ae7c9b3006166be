#!/usr/bin/env python3
"""Phases of the C3 train step on the MAIN stream, un-profiled (HIP events at the phase boundaries; a kernel trace adds
~7 us per dispatch = ~8 ms per step and turns overlapped stretches into idle ones): GPU-side duration of every phase,
the host's time at the same marker (how far the host runs ahead), medians over the measured steps.

    python tools/phase_probe.py [--index 5] [--steps 12] [--batch 32]"""
import argparse
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--index', type=int, default=5)
ap.add_argument('--steps', type=int, default=12)
ap.add_argument('--batch', type=int, default=32)
a = ap.parse_args()
from histoGAN import Trainer  # noqa: E402

tmp = tempfile.mkdtemp()
tr = Trainer('p', tmp + '/r', tmp + '/m', 256, 16, batch_size=a.batch, hist_insz=150, hist_resizing='interpolation')
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src()
for i in range(8):
    tr.train()
for _ in range(3):
    tr.steps = a.index
    tr.train()
torch.cuda.synchronize()
import histogan_amd.nets as HN  # noqa: E402


def hook(name, any_stream=False):   # marks inside the networks: on the step's main stream (the side-stream forward overlaps
    # other phases), or -- backward nodes, which the engine runs one after the other -- on whatever stream they run
    if tr.__dict__.get('phase_events') is not None and (any_stream or torch.cuda.current_stream() == torch.cuda.default_stream()):
        tr._mark(name)


HN.PHASE_HOOK = hook
rows = []
for _ in range(a.steps):
    tr.steps = a.index
    tr.phase_events = []
    tr.train()
    rows.append(tr.phase_events)
tr.phase_events = None
torch.cuda.synchronize()
med = lambda v: sorted(v)[len(v) // 2]
names = [n for n, _, _ in rows[0]]
out = {'index': a.index, 'batch': a.batch, 'phases': []}
for i in range(1, len(names)):
    gpu = med([r[i - 1][1].elapsed_time(r[i][1]) for r in rows])
    host = med([(r[i][2] - r[i - 1][2]) * 1e3 for r in rows])
    out['phases'].append({'phase': names[i], 'gpu_ms': round(gpu, 3), 'host_ms': round(host, 3)})
out['step_gpu_ms_start_to_optimizer'] = round(med([r[0][1].elapsed_time(r[-1][1]) for r in rows]), 3)
out['step_gpu_ms_start_to_start'] = round(med([rows[j][0][1].elapsed_time(rows[j + 1][0][1]) for j in range(len(rows) - 1)]), 3)
out['host_lead_ms_at_end'] = round(med([r[0][1].elapsed_time(r[-1][1]) - (r[-1][2] - r[0][2]) * 1e3 for r in rows]), 3)
print(json.dumps(out))
