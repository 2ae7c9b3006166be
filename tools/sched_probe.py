#!/usr/bin/env python3
"""Plain-step GPU time at C3 (256^2, capacity 16, batch 32) under scheduling knobs, for A/B runs:

    HG_G_STREAM_PRIO=-1 python tools/sched_probe.py            # one configuration per process (stream priorities)
    python tools/sched_probe.py --toggle D_STEP_EARLY           # a trainer module flag flipped between interleaved blocks
Prints one JSON line: median / min GPU ms between end-of-step events."""
import argparse
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=16)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--index', type=int, default=5)
ap.add_argument('--toggle', default='')
a = ap.parse_args()
from histoGAN import Trainer  # noqa: E402
from histogan_amd import trainer as T  # noqa: E402

tmp = tempfile.mkdtemp()
tr = Trainer('ab', tmp + '/r', tmp + '/m', 256, 16, batch_size=32, hist_insz=150, hist_resizing='interpolation')
tr.run_evaluate = tr.run_save = False
tr.graph_mode = '0'
tr.set_synthetic_data_src()
tr.init_GAN()
for i in range(8):
    tr.train()
vals = [True, False] if a.toggle else [None]
res = {str(v): [] for v in vals}
for r in range(a.rounds):
    for v in vals:
        if a.toggle:
            mod, _, attr = a.toggle.rpartition(':')          # 'D_STEP_EARLY' (trainer) or 'histogan_amd.conv:DIRECT_DEMOD'
            setattr(__import__(mod, fromlist=['x']) if mod else T, attr, v)
        for _ in range(2):
            tr.steps = a.index
            tr.train()
        torch.cuda.synchronize()
        tr.keep_step_events, tr.step_events = True, []
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(a.steps):
            tr.steps = a.index
            tr.train()
        torch.cuda.synchronize()
        evs = [e0] + [e for _, e in tr.step_events]
        res[str(v)] += [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
        tr.keep_step_events = False
med = lambda x: sorted(x)[len(x) // 2]
env = {k: os.environ[k] for k in os.environ if k.startswith('HG_')}
print(json.dumps(dict(env=env, toggle=a.toggle, **{k: dict(median=round(med(x), 3), min=round(min(x), 3), n=len(x)) for k, x in res.items()})))
