#!/usr/bin/env python3
"""hipGraph replay of the plain train step vs the same static-input step run eagerly (HG_GRAPH=2) vs the ordinary
eager step (HG_GRAPH=0): step times, host enqueue time, and -- same seed -- the parameter checksums after N steps.
    HG_GRAPH=1 python tools/graph_probe.py [steps] [batch] [size] [capacity]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histoGAN import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 16
import random
random.seed(0)
torch.manual_seed(0)
tr = Trainer('gp', '/tmp/gp_results', '/tmp/gp_models', size, cap, batch_size=batch, hist_bin=64, hist_insz=150,
             hist_resizing='interpolation')
tr.run_evaluate = tr.run_save = False
tr.set_synthetic_data_src()
tr.init_GAN()
host, total, kinds = [], [], []
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp, pl = tr.steps % 4 == 0, tr.steps % 32 == 0
    graphed = tr._graph_eligible(gp, pl)
    tr.train(alpha=2)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    total.append((t2 - t0) * 1e3); kinds.append('graph' if graphed else ('gp' if gp else 'plain'))
    host.append(tr.host_enqueue_ms)
    print(f'step {i:3d} {kinds[-1]:6s} {total[-1]:8.2f} ms  host {tr.host_enqueue_ms:6.2f} ms  D {tr.d_loss:8.4f} G {tr.g_loss:8.4f} H {tr.h_loss:8.4f}', flush=True)
for k in ('graph', 'plain', 'gp'):
    v = [t for t, kk in zip(total[8:], kinds[8:]) if kk == k]
    if v:
        hv = [t for t, kk in zip(host[8:], kinds[8:]) if kk == k]
        print(f'{k:6s}: n={len(v)} mean {sum(v)/len(v):.2f} ms  min {min(v):.2f}   host enqueue mean {sum(hv)/len(hv):.2f} ms')
g = tr.GAN
print('checksum G %.10e D %.10e' % (float(g._flat_g.data.double().sum()), float(g._flat_d.data.double().sum())))
print('abs-checksum G %.10e D %.10e' % (float(g._flat_g.data.double().abs().sum()), float(g._flat_d.data.double().abs().sum())))
print('mode HG_GRAPH=%s failed=%s' % (os.environ.get('HG_GRAPH', '1'), getattr(tr, '_graph_failed', False)))
