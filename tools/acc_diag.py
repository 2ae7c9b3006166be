#!/usr/bin/env python3
"""Where does the forward's distance to the fp64 evaluation come from?  (debug aid, GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from bigcases import load_big
from oracle import rgbuv_hist as O
from histogram_classes.RGBuvHistBlock import RGBuvHistBlock

dev = torch.device('cuda:0')
for name in sys.argv[1:] or ['c2_2x256_uniform']:
    g = load_big(name)
    kw = g['spec']['kw']
    x = g['x']
    if g['spec'].get('relu'): x = torch.relu(x)
    truth = O.rgbuv_hist(x, truth=True, **kw).numpy()
    ours = RGBuvHistBlock(device='cuda', **kw)(x.to(dev)).cpu().numpy().astype(np.float64)
    ref = g['hist'].astype(np.float64)
    m = np.abs(truth).max()
    for lab, a in (('ours', ours), ('ref ', ref)):
        e = (a - truth)
        rel = e / np.maximum(truth, 1e-30)
        big = truth > 0.1 * m
        i = np.unravel_index(np.argmax(np.abs(e)), e.shape)
        print(f'{name} {lab}: max|e|/max {np.abs(e).max()/m:.2e} rms(e)/max {np.sqrt((e**2).mean())/m:.2e} '
              f'mean signed rel err (all bins) {rel.mean():+.2e}  (bins > 0.1 max: mean {rel[big].mean():+.2e} rms {np.sqrt((rel[big]**2).mean()):.2e}) '
              f'sum-1 {a.reshape(a.shape[0], -1).sum(1) - 1}  argmax err at {i} truth there {truth[i]/m:.3f} of max')
