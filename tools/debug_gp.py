#!/usr/bin/env python3
"""Debug: discriminator gradients of a gradient-penalty step at C3 (B = 2) through the Trainer vs the fp64 oracle, with the
direct side-stream weight gradients on and off, fused lrelu-backward on and off."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch  # noqa: E402

from oracle_step import ReplayRng, oracle_train_step  # noqa: E402
from oracle import rgbuv_hist as OH  # noqa: E402
import histogan_amd.conv as C  # noqa: E402
from histoGAN import Trainer  # noqa: E402

S_, CAP, HB, LAT, B = 256, 16, 64, 512, 2
dev = torch.device('cuda:0')


def run(tag, side, fused_lrelu=True, conv_add=True, overlap=True, packcache=True):
    import histogan_amd.trainer as T
    T.G_OVERLAP = overlap
    C.SIDE_WGRAD = side
    torch.manual_seed(31)
    tmp = tempfile.mkdtemp()
    tr = Trainer('dbg', tmp + '/r', tmp + '/m', S_, CAP, batch_size=B, lr=2e-4, hist_bin=HB, hist_insz=150,
                 hist_resizing='interpolation', mixed_prob=1.1)
    tr.graph_mode = '0'
    tr.run_evaluate = tr.run_save = False
    tr.init_GAN()
    if not packcache:
        C.enable_pack_cache(None)
    GAN = tr.GAN
    L = GAN.G.num_layers
    sd0 = {k: v.detach().clone() for k, v in GAN.state_dict().items()}
    gen = torch.Generator().manual_seed(6)
    batches = []
    for _ in range(2):
        img = torch.rand(B, 3, S_, S_, generator=gen)
        hist = OH.rgbuv_hist(torch.rand(B, 3, S_, S_, generator=gen), h=HB)
        batches.append({'images': img.to(dev), 'histograms': hist.to(dev)})
    tr.loader = iter(batches)
    tr.rng = ReplayRng(dev, B, L, LAT, S_, 78, tt=2)
    tr.steps = 4
    if not fused_lrelu:
        orig = C.lrelu_bwd_channel_sum
        C.lrelu_bwd_channel_sum = lambda g, out, s, w=True: (torch.ops.aten.leaky_relu_backward(g, out, s, True),
                                                             (torch.ops.aten.leaky_relu_backward(g, out, s, True).sum(dim=(0, 2, 3)) if w else None))
    tr.train(alpha=2.0)
    torch.cuda.synchronize()
    if not fused_lrelu:
        C.lrelu_bwd_channel_sum = orig
    grads, off = {}, 0
    for prm in GAN._flat_d.params:
        n = prm.numel()
        name = next(k for k, v in GAN.D.named_parameters() if v is prm)
        grads[name] = GAN._flat_d.grad[off:off + n].view(prm.shape).clone()
        off += n
    return sd0, batches, L, grads


sd0, batches, L, gA = run('direct', True)
_, _, _, gB = run('no_overlap', True, overlap=False)
_, _, _, gC = run('no_packcache', True, packcache=False)
truth = oracle_train_step(sd0, batches, ReplayRng(dev, B, L, LAT, S_, 78, tt=2, dtype=torch.float64), L, HB, 2.0, 2e-4, True,
                          False, optimizer=False)
rel = lambda a, t: float((a.double() - t).abs().max() / t.abs().max().clamp_min(1e-300))
rows = []
for name in gA:
    t = truth['grads'][('D', name)]
    rows.append((rel(gA[name], t), rel(gB[name], t), rel(gC[name], t), name, tuple(t.shape)))
rows.sort(reverse=True)
print('direct        no G overlap  no pack cache   name')
for r in rows[:14]:
    print(f'{r[0]:.3e}    {r[1]:.3e}    {r[2]:.3e}    {r[3]} {r[4]}')
w = rows[0][3]
t = truth['grads'][('D', w)]
e = (gA[w].double() - t).abs()
thr = 0.1 * e.max()
idx = (e > thr).nonzero()
print('worst tensor', w, 'elements above 10% of max error:', idx.shape[0], 'of', e.numel())
print(idx[:20].tolist())
print('ratio ours/truth at worst elements:', [(float(gA[w][tuple(i)]), float(t[tuple(i)])) for i in idx[:6]])
print('direct vs autograd max abs diff on worst tensor', float((gA[w] - gB[w]).abs().max()), 'scale', float(t.abs().max()))
