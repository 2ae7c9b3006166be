#!/usr/bin/env python3
"""Same-process A/B of the train step's execution modes at C3 (256^2, capacity 16, batch 32): GPU-side time between the
end-of-step events, plain steps only (step index pinned), modes interleaved so box / clock drift hits all alike.

    python tools/ab_step.py [--steps 16] [--rounds 2] [--index 5]
modes: eager + blocking read-back (round 2's step), eager + deferred read-back, hipGraph replay + deferred read-back."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=16)
ap.add_argument('--rounds', type=int, default=2)
ap.add_argument('--index', type=int, default=5)
ap.add_argument('--batch', type=int, default=32)
a = ap.parse_args()
from histoGAN import Trainer  # noqa: E402

tmp = tempfile.mkdtemp()
tr = Trainer('ab', tmp + '/r', tmp + '/m', 256, 16, batch_size=a.batch, hist_insz=150, hist_resizing='interpolation')
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src()
tr.init_GAN()
for i in range(8):                       # warm-up incl. a gradient-penalty / path-length step, allocator growth
    tr.train()
MODES = [('eager_sync', dict(lazy_stats=False, graph_mode='0')), ('eager_lazy', dict(lazy_stats=True, graph_mode='0')),
         ('graph_lazy', dict(lazy_stats=True, graph_mode='1'))]
res = {m: [] for m, _ in MODES}
wall = {m: [] for m, _ in MODES}
for r in range(a.rounds):
    for name, kw in MODES:
        for k, v in kw.items():
            setattr(tr, k, v)
        for _ in range(3):               # settle (graph capture on first use)
            tr.steps = a.index if name != 'graph_lazy' else max(a.index, 9)
            tr.train()
        torch.cuda.synchronize()
        tr.keep_step_events, tr.step_events = True, []
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tr.steps = a.index if name != 'graph_lazy' else max(a.index, 9)
            tr.train()
        torch.cuda.synchronize()
        wall[name].append((time.perf_counter() - t0) / a.steps * 1e3)
        evs = [e0] + [e for _, e in tr.step_events]
        res[name] += [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
        tr.keep_step_events = False
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps({m: dict(gpu_ms_median=med(v), gpu_ms_min=min(v), wall_ms_per_step=wall[m], n=len(v),
                          host_enqueue_ms=tr.host_enqueue_ms if m == MODES[-1][0] else None) for m, v in res.items()}))
