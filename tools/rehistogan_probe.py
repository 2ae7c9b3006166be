#!/usr/bin/env python3
"""Time the ReHistoGAN train step (recoloringTrainer.train) on synthetic data: ms/step and images/s.

    python tools/rehistogan_probe.py [--size 256] [--cap 16] [--batch 8] [--steps 16] [--skip] [--internal] [--var]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--cap', type=int, default=16)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--skip', action='store_true')
    ap.add_argument('--internal', action='store_true')
    ap.add_argument('--var', action='store_true')
    ap.add_argument('--rec', default='laplacian')
    a = ap.parse_args()
    from ReHistoGAN import recoloringTrainer
    tmp = tempfile.mkdtemp()
    tr = recoloringTrainer('probe', os.path.join(tmp, 'r'), os.path.join(tmp, 'm'), a.size, a.cap, batch_size=a.batch,
                           skip_conn_to_GAN=a.skip, internal_hist=a.internal, variance_loss=a.var, rec_loss=a.rec)
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    for _ in range(a.warmup):
        tr.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    n = lambda m: sum(p.numel() for p in m.parameters())
    print(json.dumps(dict(workload=f'rehistogan train {a.size}^2 cap{a.cap} B={a.batch} skip={a.skip} '
                          f'internal={a.internal} var={a.var} rec={a.rec}', ms_per_step=round(dt * 1e3, 3),
                          images_per_s=round(a.batch / dt, 2), params=dict(ED=n(tr.GAN.ED), G=n(tr.GAN.G), H=n(tr.GAN.H),
                                                                           D=n(tr.GAN.D)),
                          losses=dict(d=tr.d_loss, g=tr.g_loss, r=tr.r_loss, h=tr.h_loss),
                          mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))))


if __name__ == '__main__':
    main()
