"""Generator(256, 512, 16) gradients vs the fp64 oracle over several input draws, with / without the chunked skinny GEMM:
against the oracle on its own LeakyReLU branches (natural) and on the branches our forward took (same-mask)."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import torch
from histoGAN import Generator
from histogan_amd import ops
from oracle import histogan_nets as N
from oracle_step import LreluMargin, LreluMasks
dev = torch.device('cuda:0')
torch.manual_seed(21)
B, S_, LAT, CAP = 2, 256, 512, 16
G = Generator(S_, LAT, network_capacity=CAP).to(dev)
with torch.no_grad():
    for blk in G.blocks:
        blk.to_noise1.weight.normal_(std=0.3); blk.to_noise2.weight.normal_(std=0.3)
        blk.to_noise1.bias.normal_(std=0.1); blk.to_noise2.bias.normal_(std=0.1)
L = G.num_layers
names = [n for n, _ in G.named_parameters()]
params = dict(G.named_parameters())
rel = lambda a, t: float((a.double() - t).abs().max() / t.abs().max().clamp_min(1e-30))
for seed in range(21, 29):
    g = torch.Generator(device='cpu').manual_seed(seed)
    styles = torch.randn(B, L - 2, LAT, generator=g).to(dev).requires_grad_(True)
    hists = torch.randn(B, 2, LAT, generator=g).to(dev).requires_grad_(True)
    noise = torch.rand(B, S_, S_, 1, generator=g).to(dev)
    go = torch.randn(B, 3, S_, S_, generator=g).to(dev)
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    st, hi = styles.detach().double().requires_grad_(True), hists.detach().double().requires_grad_(True)
    with LreluMargin() as lm:
        o = N.generator(sd, st, hi, noise.double(), L)
    t_gr = torch.autograd.grad(o, [st, hi] + [sd[n] for n in names], go.double())
    row = [seed, 'margin %.1e' % lm.value]
    for flag in (False, True):
        ops.SKINNY_SPLIT = flag
        masks, orig = [], ops.demod_noise_lrelu
        def rec(*a):
            out = orig(*a); masks.append(out.detach() > 0); return out
        ops.demod_noise_lrelu = rec
        rgb = G(styles, hists, noise)
        ops.demod_noise_lrelu = orig
        grads = torch.autograd.grad(rgb, [styles, hists] + [params[n] for n in names], go)
        nat = max(rel(a, t) for a, t in zip(grads, t_gr))
        sd2 = {k: v.detach().double().clone().requires_grad_(True) for k, v in G.state_dict().items()}
        st2, hi2 = styles.detach().double().requires_grad_(True), hists.detach().double().requires_grad_(True)
        with LreluMasks(masks) as mm:
            o2 = N.generator(sd2, st2, hi2, noise.double(), L)
        m_gr = torch.autograd.grad(o2, [st2, hi2] + [sd2[n] for n in names], go.double())
        same = max(rel(a, t) for a, t in zip(grads, m_gr))
        row.append('%s: natural %.1e same-mask %.1e flips %d (margin %.1e)' % ('split' if flag else 'plain', nat, same, mm.flips, mm.flip_margin))
    print(row)
