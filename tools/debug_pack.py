#!/usr/bin/env python3
"""Debug: batched packing (hg_conv_pack_weights_multi) of every D / G conv weight of a C3 trainer vs the single packs."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import histogan_amd.conv as C  # noqa: E402
from histoGAN import Trainer  # noqa: E402

tmp = tempfile.mkdtemp()
tr = Trainer('dbg', tmp + '/r', tmp + '/m', 256, 16, batch_size=2, hist_insz=150, hist_resizing='interpolation')
tr.init_GAN()
bad = 0
for net_name in ('D', 'G', 'GE'):
    net = getattr(tr.GAN, net_name)
    for name, p in net.named_parameters():
        if p.dim() != 4:
            continue
        for mode in (C.PACK_FWD, C.PACK_DGRAD):
            a = C.pack_weights(p, mode)
            b = C._pack_weights(p.detach(), mode)
            if not torch.equal(a, b):
                bad += 1
                d = (a - b).abs()
                idx = d.nonzero()
                print('MISMATCH', net_name, name, tuple(p.shape), 'mode', mode, 'n', idx.shape[0], 'first', idx[:6].flatten().tolist(),
                      'ptr % 16 =', p.data_ptr() % 16)
        wq = C.cached(p, 'wsq', lambda t: None)
        if wq is not None:
            ref = p.detach().pow(2).sum(dim=(2, 3))
            e = float((wq - ref).abs().max() / ref.abs().max())
            if e > 1e-5:
                print('WSQ MISMATCH', net_name, name, e)
                bad += 1
print('mismatching operands:', bad)
