#!/usr/bin/env python3
"""LeakyReLU mask flips between an fp32 evaluation and fp64 (how DESIGN.md section 0's finding was made; replaces round 3's
tools/debug_gp*.py).  A pre-activation within fp32 rounding of zero gets the other LeakyReLU slope in ANY fp32 evaluation
than in fp64; at B = 2 one such pixel moves a small weight-gradient tensor by 1e-2 through the gradient penalty's
second-order terms.  The parity tests therefore pick input data whose closest pre-activation is clear of zero
(tests/oracle_step.lrelu_margin) or compare on the branches the forward took (oracle_step.LreluMasks).

    python tools/lrelu_margin_probe.py [data_seed]        # on the GPU box

Same set-up as tests/test_c3_parity_gpu.py (256^2, capacity 16, B = 2, seed 31): compares the sign of every LeakyReLU output
of the discriminator (our fp32 kernels vs fp64 torch) on the batch of `data_seed` and reports the fp64 pre-activation
magnitude at every flipped element relative to the layer's largest."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import rgbuv_hist as OH  # noqa: E402
from histogan_amd.conv import conv2d_lrelu  # noqa: E402
from histoGAN import Trainer  # noqa: E402

S_, CAP, HB, B = 256, 16, 64, 2
dev = torch.device('cuda:0')
torch.manual_seed(31)
tmp = tempfile.mkdtemp()
tr = Trainer('dbg', tmp + '/r', tmp + '/m', S_, CAP, batch_size=B, lr=2e-4, hist_bin=HB, hist_insz=150,
             hist_resizing='interpolation', mixed_prob=1.1)
tr.init_GAN()
D = tr.GAN.D
gen = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
img = torch.rand(B, 3, S_, S_, generator=gen).to(dev)
sd = {k: v.detach().double() for k, v in D.state_dict().items()}
x32, x64 = img.clone(), img.double()
for i, blk in enumerate(D.blocks):
    p = f'blocks.{i}.'
    h32 = conv2d_lrelu(x32, blk.net[0].weight, blk.net[0].bias, 0.2)
    pre64 = F.conv2d(x64, sd[p + 'net.0.weight'], sd[p + 'net.0.bias'], padding=1)
    h64 = F.leaky_relu(pre64, 0.2)
    for tag, a, pre in (('net.0', h32, pre64),):
        flip = (a > 0) != (pre > 0)
        n = int(flip.sum())
        print(f'block {i} {tag}: {n} sign flips of {a.numel()}', end='')
        if n:
            idx = flip.nonzero()
            print('  at', idx[:4].tolist(), ' |pre64| / max =', [float(pre[tuple(j)].abs() / pre.abs().max()) for j in idx[:4]], end='')
        print()
    g32 = conv2d_lrelu(h32, blk.net[2].weight, blk.net[2].bias, 0.2)
    pre64b = F.conv2d(h64, sd[p + 'net.2.weight'], sd[p + 'net.2.bias'], padding=1)
    flip = (g32 > 0) != (pre64b > 0)
    n = int(flip.sum())
    print(f'block {i} net.2: {n} sign flips of {g32.numel()}', end='')
    if n:
        idx = flip.nonzero()
        print('  at', idx[:4].tolist(), ' |pre64| / max =', [float(pre64b[tuple(j)].abs() / pre64b.abs().max()) for j in idx[:4]], end='')
    print()
    x64 = F.leaky_relu(pre64b, 0.2) + F.conv2d(x64, sd[p + 'conv_res.weight'], sd[p + 'conv_res.bias'])
    if blk.downsample is not None:
        x64 = F.conv2d(x64, sd[p + 'downsample.weight'], sd[p + 'downsample.bias'], padding=1, stride=2)
    x32 = x64.float()          # continue from the fp64 activations: flips are counted layer by layer, not accumulated
