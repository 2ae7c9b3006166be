"""GPU parity of the generator kernels / network modules / optimizer / train step against
(a) golden vectors from the reference's own classes and (b) the functional oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN_DIR, relmax

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def g():
    z = np.load(os.path.join(GOLDEN_DIR, 'nets_small.npz'))
    return {k: z[k] for k in z.files}


def sd_of(g, prefix, dev):
    return {k[len(prefix) + 1:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith(prefix + '/')}


def T(a, dev, grad=False):
    t = torch.from_numpy(np.asarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


# ---- kernels vs plain torch ------------------------------------------------------------------
@pytest.mark.parametrize('shape,up', [((2, 5, 4, 4), False), ((2, 5, 4, 4), True), ((3, 7, 16, 16), True),
                                      ((1, 3, 32, 32), True), ((2, 4, 64, 64), False), ((2, 3, 2, 2), True)])
def test_modulate_matches_torch(shape, up, gpu_device):
    from histogan_amd import ops
    torch.manual_seed(1)
    x = torch.randn(*shape, device=gpu_device, requires_grad=True)
    s = torch.randn(shape[0], shape[1], device=gpu_device, requires_grad=True)
    ref = (F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) if up else x) * (s + 1)[:, :, None, None]
    out = ops.modulate(x, s, up)
    assert relmax(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-6
    go = torch.randn_like(ref)
    gr = torch.autograd.grad(ref, (x, s), go)
    gm = torch.autograd.grad(out, (x, s), go)
    for a, b in zip(gm, gr):
        assert relmax(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5


def test_upsample2x_matches_torch(gpu_device):
    from histogan_amd import ops
    x = torch.randn(2, 3, 8, 8, device=gpu_device, requires_grad=True)
    ref = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    out = ops.upsample2x(x)
    assert relmax(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-6
    go = torch.randn_like(ref)
    assert relmax(torch.autograd.grad(out, x, go)[0].cpu().numpy(),
                  torch.autograd.grad(ref, x, go)[0].cpu().numpy()) <= 1e-5


@pytest.mark.parametrize('H,O,demod', [(4, 6, True), (16, 5, True), (32, 3, False)])
def test_demod_noise_lrelu_matches_torch(H, O, demod, gpu_device):
    from histogan_amd import ops
    torch.manual_seed(2)
    B, S = 2, 64
    conv = torch.randn(B, O, H, H, device=gpu_device, requires_grad=True)
    d = (torch.rand(B, O, device=gpu_device) + 0.5).requires_grad_(True) if demod else None
    inoise = torch.rand(B, S, S, 1, device=gpu_device)
    lin = torch.nn.Linear(1, O).to(gpu_device)
    noise = lin(inoise[:, :H, :H, :]).permute(0, 3, 2, 1)          # the reference's expression (:465-467)
    ref = F.leaky_relu((conv * d[:, :, None, None] if demod else conv) + noise, 0.2)
    out = ops.demod_noise_lrelu(conv, d, inoise[..., 0].transpose(1, 2).contiguous(), lin.weight, lin.bias)
    assert relmax(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-6
    go = torch.randn_like(ref)
    ins = [conv, lin.weight, lin.bias] + ([d] if demod else [])
    gr = torch.autograd.grad(ref, ins, go)
    gm = torch.autograd.grad(out, ins, go)
    for a, b in zip(gm, gr):
        assert relmax(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5


# ---- modules vs goldens of the reference classes ---------------------------------------------
@pytest.mark.parametrize('tag,k,demod', [('c3', 3, True), ('c1', 1, False)])
def test_conv2dmod_golden(g, tag, k, demod, gpu_device):
    from histoGAN import Conv2DMod
    w = g[f'{tag}/weight']
    conv = Conv2DMod(w.shape[1], w.shape[0], k, demod=demod).to(gpu_device)
    with torch.no_grad():
        conv.weight.copy_(T(w, gpu_device))
    x, y = T(g[f'{tag}/x'], gpu_device, True), T(g[f'{tag}/y'], gpu_device, True)
    o = conv(x, y)
    assert relmax(o.detach().cpu().numpy(), g[f'{tag}/out']) <= 1e-5
    gx, gy, gw = torch.autograd.grad(o, (x, y, conv.weight), T(g[f'{tag}/go'], gpu_device))
    for a, name in ((gx, 'gx'), (gy, 'gy'), (gw, 'gw')):
        assert relmax(a.cpu().numpy(), g[f'{tag}/{name}']) <= 1e-4


def test_generator_golden(g, gpu_device):
    from histoGAN import Generator
    S_, CAP, LAT, HB, B, L = [int(v) for v in g['meta']]
    G = Generator(S_, LAT, network_capacity=CAP).to(gpu_device)
    G.load_state_dict(sd_of(g, 'G', gpu_device))
    styles, hists = T(g['g_styles'], gpu_device, True), T(g['g_hists'], gpu_device, True)
    rgb = G(styles, hists, T(g['g_noise'], gpu_device))
    assert relmax(rgb.detach().cpu().numpy(), g['g_rgb']) <= 1e-5
    names = [k[len('g_grad/'):] for k in g if k.startswith('g_grad/')]
    params = dict(G.named_parameters())
    grads = torch.autograd.grad(rgb, [styles, hists] + [params[n] for n in names], T(g['g_go'], gpu_device))
    assert relmax(grads[0].cpu().numpy(), g['g_grad_styles']) <= 1e-4
    assert relmax(grads[1].cpu().numpy(), g['g_grad_hists']) <= 1e-4
    for n, gr in zip(names, grads[2:]):
        assert relmax(gr.cpu().numpy(), g[f'g_grad/{n}']) <= 1e-4, n


def test_discriminator_and_gp_golden(g, gpu_device):
    from histoGAN import Discriminator
    from histoGAN.histoGAN import gradient_penalty
    S_, CAP = int(g['meta'][0]), int(g['meta'][1])
    D = Discriminator(S_, network_capacity=CAP).to(gpu_device)
    D.load_state_dict(sd_of(g, 'D', gpu_device))
    img = T(g['d_img'], gpu_device, True)
    logits, q = D(img)
    assert relmax(logits.detach().cpu().numpy(), g['d_logits']) <= 1e-5
    gp = gradient_penalty(img, logits)
    assert abs(float(gp) - float(g['d_gp'])) <= 1e-4 * max(1.0, abs(float(g['d_gp'])))
    loss = torch.relu(1 + logits).mean() + gp
    names = [k[len('d_grad/'):] for k in g if k.startswith('d_grad/')]
    params = dict(D.named_parameters())
    for n, gr in zip(names, torch.autograd.grad(loss, [params[n] for n in names])):
        assert relmax(gr.cpu().numpy(), g[f'd_grad/{n}']) <= 1e-4, n


def test_vectorizers_golden(g, gpu_device):
    from histoGAN import HistVectorizer
    from histoGAN.histoGAN import StyleVectorizer
    S_, CAP, LAT, HB, B, L = [int(v) for v in g['meta']]
    sv = StyleVectorizer(LAT, 3).to(gpu_device); sv.load_state_dict(sd_of(g, 'S', gpu_device))
    hv = HistVectorizer(HB, LAT, 3).to(gpu_device); hv.load_state_dict(sd_of(g, 'H', gpu_device))
    assert relmax(sv(T(g['z'], gpu_device)).detach().cpu().numpy(), g['w']) <= 1e-5
    assert relmax(hv(T(g['hist'], gpu_device)).detach().cpu().numpy(), g['hw']) <= 1e-5


# ---- optimizer --------------------------------------------------------------------------------
def test_diffgrad_and_ema_match_oracle(gpu_device):
    from histogan_amd.optim import DiffGrad, FlatParams, ema_update
    from oracle import histogan_nets as N
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(s, device=gpu_device)) for s in ((7, 5), (33,), (4, 3, 3, 3))]
    ref = [p.detach().cpu().clone() for p in ps]
    states = [dict(step=0, exp_avg=torch.zeros_like(r), exp_avg_sq=torch.zeros_like(r),
                   previous_grad=torch.zeros_like(r)) for r in ref]
    opt = DiffGrad(ps, lr=2e-4, betas=(0.5, 0.9))
    for it in range(5):
        opt.zero_grad()
        gs = [torch.randn_like(r) * (0.1 + it) for r in ref]
        for p, gr in zip(ps, gs):
            p.grad = gr.to(gpu_device)          # as autograd leaves it; DiffGrad.step gathers into the flat buffer
        opt.step()
        for r, gr, stt in zip(ref, gs, states):
            N.diffgrad_step(r, gr, stt, lr=2e-4, betas=(0.5, 0.9))
    for p, r in zip(ps, ref):
        assert relmax(p.detach().cpu().numpy(), r.numpy()) <= 1e-6
    ma = [torch.nn.Parameter(torch.randn_like(p)) for p in ps]
    ma_ref = [m.detach().cpu().clone() for m in ma]
    fm = FlatParams(ma, with_grad=False)
    ema_update(fm, opt.flat, 0.995)
    for m, mr, p in zip(ma, ma_ref, ps):
        assert relmax(m.detach().cpu().numpy(), (mr * 0.995 + 0.005 * p.detach().cpu()).numpy()) <= 1e-6


# ---- whole train step vs the oracle ------------------------------------------------------------
def test_train_step_matches_oracle(gpu_device, tmp_path):
    """One Trainer.train() step at step 0 (gradient penalty AND path-length regulariser active) against the same step
    evaluated with oracle/ (functional nets + oracle histogram + oracle DiffGrad, tests/oracle_step.py) on the CPU in
    fp32 (the reference's numerics) and in fp64 (truth).  The C3-shape version is tests/test_c3_parity_gpu.py."""
    from histoGAN import Trainer
    from oracle import rgbuv_hist as OH
    from oracle_step import ReplayRng, oracle_train_step
    torch.manual_seed(11)
    S_, CAP, B, HB, ALPHA, LR = 32, 4, 2, 16, 2.0, 2e-4
    tr = Trainer('t', tmp_path / 'r', tmp_path / 'm', S_, CAP, batch_size=B, lr=LR, hist_bin=HB, hist_insz=150,
                 hist_resizing='interpolation', mixed_prob=1.1)
    tr.run_evaluate = tr.run_save = False
    tr.init_GAN()
    GAN = tr.GAN
    with torch.no_grad():   # exercise the noise path (zero-initialised in the reference)
        for blk in GAN.G.blocks:
            blk.to_noise1.weight.normal_(std=0.3); blk.to_noise2.weight.normal_(std=0.3)
    L, LAT = GAN.G.num_layers, GAN.G.latent_dim
    sd0 = {k: v.detach().cpu().clone() for k, v in GAN.state_dict().items()}
    gen = torch.Generator().manual_seed(5)
    batches = []
    for _ in range(2):
        img = torch.rand(B, 3, S_, S_, generator=gen)
        hist = OH.rgbuv_hist(torch.rand(B, 3, S_, S_, generator=gen), h=HB)
        batches.append({'images': img, 'histograms': hist})
    tr.loader = iter([{k: v.to(gpu_device) for k, v in b.items()} for b in batches])
    tr.rng = ReplayRng(gpu_device, B, L, LAT, S_, 77)
    tr.train(alpha=ALPHA)
    new = {k: v.detach().cpu() for k, v in GAN.state_dict().items()}

    cpu = torch.device('cpu')
    ref = oracle_train_step(sd0, batches, ReplayRng(cpu, B, L, LAT, S_, 77), L, HB, ALPHA, LR, True, True)
    truth = oracle_train_step(sd0, batches, ReplayRng(cpu, B, L, LAT, S_, 77, dtype=torch.float64), L, HB, ALPHA, LR,
                              True, True)
    assert abs(tr.d_loss - ref['d_loss']) <= 1e-4
    assert abs(tr.g_loss - ref['g_loss']) <= 1e-4
    assert abs(tr.h_loss - ref['h_loss']) <= 1e-4
    assert abs(tr.last_gp_loss - ref['gp']) <= 1e-4 * max(1.0, abs(ref['gp']))
    # Generator-side gradients (still in the flat buffer; zeroed at the start of the next step), SURVEY 8(c) criterion:
    # our distance to the fp64 evaluation within 2x the fp32 reference's own.  (Round 2 asserted 3e-3 against the fp32
    # oracle and explained it by the histogram's 1 / (x + 1e-6) gradient at relu / clamp-edge pixels amplifying 1e-7
    # differences of the generator output; measured here, both fp32 evaluations sit equally far from the truth.)
    gkeys = [pk for pk in truth['grads'] if pk[0] != 'D']
    tn = torch.cat([truth['grads'][pk].flatten() for pk in gkeys]).norm()
    mine = {pk: dict(getattr(GAN, pk[0]).named_parameters())[pk[1]].grad.detach().cpu().double() for pk in gkeys}
    d_ours = float(torch.cat([(mine[pk] - truth['grads'][pk]).flatten() for pk in gkeys]).norm() / tn)
    d_ref = float(torch.cat([(ref['grads'][pk].double() - truth['grads'][pk]).flatten() for pk in gkeys]).norm() / tn)
    assert d_ours <= 2 * d_ref + 1e-7, (d_ours, d_ref)
    for pk in gkeys:
        e_o = relmax(mine[pk].numpy(), truth['grads'][pk].numpy())
        e_r = relmax(ref['grads'][pk].numpy(), truth['grads'][pk].numpy())
        assert e_o <= max(1e-4, 3 * e_r), (pk, e_o, e_r)
    # parameters after the step.  The first DiffGrad step is ~ lr*sigmoid(|g|)*g/(|g|+3e-8): compare the
    # deltas where the gradient is not rounding noise (elsewhere the sign itself is ill-conditioned)
    for (p, k), v in ref['params'].items():
        gr = ref['grads'][(p, k)].numpy()
        mask = np.abs(gr) > 5e-2 * np.abs(gr).max()   # well above the gradient tolerance: sign is certain
        dn = (new[f'{p}.{k}'] - sd0[f'{p}.{k}']).numpy()
        do = (v - sd0[f'{p}.{k}']).numpy()
        assert np.max(np.abs(dn - do)[mask]) <= 0.02 * LR, (p, k)


def test_plain_step_d_phase_matches_oracle(gpu_device, tmp_path):
    """Step 1 -- a PLAIN step: no gradient penalty, ONE discriminator pass over [fake; real] (trainer._cat_batches, the
    fused LeakyReLU-backward / bias-gradient path at twice the batch) -- against the oracle's two-pass D phase
    (reference histoGAN/histoGAN.py:889-932) on the CPU in fp64 / fp32, with the hinge ACTIVE on both halves: the logit
    layer is scaled down so that |logit| < 1 (at the kaiming initialisation relu(1 + real) = relu(1 - fake) = 0 for most
    samples and the D gradients would be compared as 0 == 0).  Discriminator gradients 1e-4 per tensor, loss 1e-4,
    masked parameter deltas after DiffGrad."""
    from histoGAN import Trainer
    from oracle import rgbuv_hist as OH
    from oracle_step import ReplayRng, oracle_train_step
    torch.manual_seed(12)
    S_, CAP, B, HB, ALPHA, LR = 32, 4, 4, 16, 2.0, 2e-4
    tr = Trainer('t', tmp_path / 'r', tmp_path / 'm', S_, CAP, batch_size=B, lr=LR, hist_bin=HB, hist_insz=150,
                 hist_resizing='interpolation', mixed_prob=1.1)
    tr.run_evaluate = tr.run_save = False
    tr.graph_mode = '0'
    tr.init_GAN()
    GAN = tr.GAN
    with torch.no_grad():
        for blk in GAN.G.blocks:
            blk.to_noise1.weight.normal_(std=0.3); blk.to_noise2.weight.normal_(std=0.3)
        GAN.D.to_logit.weight.mul_(2e-3); GAN.D.to_logit.bias.zero_()
    L, LAT = GAN.G.num_layers, GAN.G.latent_dim
    sd0 = {k: v.detach().cpu().clone() for k, v in GAN.state_dict().items()}
    gen = torch.Generator().manual_seed(6)
    batches = []
    for _ in range(2):
        img = torch.rand(B, 3, S_, S_, generator=gen)
        hist = OH.rgbuv_hist(torch.rand(B, 3, S_, S_, generator=gen), h=HB)
        batches.append({'images': img, 'histograms': hist})
    tr.loader = iter([{k: v.to(gpu_device) for k, v in b.items()} for b in batches])
    tr.rng = ReplayRng(gpu_device, B, L, LAT, S_, 79)
    tr.steps = 1
    tr.train(alpha=ALPHA)
    new = {k: v.detach().cpu() for k, v in GAN.state_dict().items()}
    cpu = torch.device('cpu')
    ref = oracle_train_step(sd0, batches, ReplayRng(cpu, B, L, LAT, S_, 79), L, HB, ALPHA, LR, False, False)
    truth = oracle_train_step(sd0, batches, ReplayRng(cpu, B, L, LAT, S_, 79, dtype=torch.float64), L, HB, ALPHA, LR,
                              False, False, split_d=True)
    assert truth['d_loss'] > 0.1, truth['d_loss']            # non-vacuous
    assert abs(tr.d_loss - truth['d_loss']) <= 1e-4
    off = 0
    bias_scale = max(float(t.abs().max()) for pk, t in truth['grads'].items() if pk[0] == 'D' and pk[1].endswith('bias'))
    for prm in GAN._flat_d.params:          # the D phase's gradients are still in D's flat gradient buffer
        n = prm.numel()
        name = next(k for k, v in GAN.D.named_parameters() if v is prm)
        mine = GAN._flat_d.grad[off:off + n].view(prm.shape).detach().cpu().double()
        t = truth['grads'][('D', name)]
        # (every sample inside the hinge: the logit gradients +-1/2B sum to zero, so the real and the fake half of a bias
        # gradient nearly cancel -- identically for to_logit.bias and the last block's conv_res.bias; bias gradients are
        # held to their un-cancelled magnitude, oracle_step: d_scale)
        den = float(t.abs().max())
        if name.endswith('bias'):
            den = max(den, truth['d_scale'][name], 1e-3 * bias_scale)
        else:
            assert den > 0, name
        e_o = float((mine - t).abs().max()) / den
        e_r = float((ref['grads'][('D', name)].double() - t).abs().max()) / den
        assert e_o <= max(1e-4, 3 * e_r), (name, e_o, e_r)
        off += n
    for k, v in truth['params_d'].items():
        gr = truth['grads'][('D', k)].numpy()
        if k.endswith('bias') and np.abs(gr).max() < 1e-3 * bias_scale:
            continue
        mask = np.abs(gr) > 5e-2 * np.abs(gr).max()
        dn = (new[f'D.{k}'].double() - sd0[f'D.{k}'].double()).numpy()
        do = (v - sd0[f'D.{k}'].double()).numpy()
        assert np.max(np.abs(dn - do)[mask]) <= 0.02 * LR, k


@pytest.mark.parametrize('shape', [(3, 5, 4, 4), (2, 16, 64, 64), (7, 3, 5, 9), (1, 1, 1, 1), (4, 130, 8, 8)])
def test_channel_sum_matches_torch(shape, gpu_device):
    from histogan_amd import ops
    torch.manual_seed(5)
    g = torch.randn(*shape, device=gpu_device)
    ref = g.double().sum(dim=(0, 2, 3))
    out = ops.channel_sum(g)
    assert relmax(out.cpu().numpy(), ref.cpu().numpy()) <= 1e-6
    assert torch.equal(out, ops.channel_sum(g))     # deterministic


@pytest.mark.parametrize('K,N,H,up,demod,act', [(6, 10, 8, False, True, True), (5, 7, 4, True, True, True),
                                                 (16, 3, 16, False, False, False), (40, 70, 16, False, True, True),
                                                 (8, 8, 32, True, True, True)])
def test_modconv_stage_matches_unfused(K, N, H, up, demod, act, gpu_device):
    """hg_modconv2d_fwd (modulation + conv + demodulation + noise + lrelu in one launch) and its backward against the
    reference expression of Conv2DMod.forward + GeneratorBlock noise/activation (histoGAN/histoGAN.py:420-440, 465-476)
    evaluated in fp64 with per-sample weights and a grouped convolution, as the reference does."""
    from histogan_amd import ops
    torch.manual_seed(7)
    B, S, k = 3, 64, 3 if act else 1
    x = torch.randn(B, K, H, H, device=gpu_device, requires_grad=True)
    y = (0.3 * torch.randn(B, K, device=gpu_device)).requires_grad_(True)
    w = (torch.randn(N, K, k, k, device=gpu_device) / (K * k * k) ** 0.5).requires_grad_(True)
    inoise = torch.rand(B, S, S, 1, device=gpu_device)
    lin = torch.nn.Linear(1, N).to(gpu_device)
    Ho = 2 * H if up else H

    def reference(x, y, w, lw, lb):
        xd = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) if up else x
        wts = w[None] * (y[:, None, :, None, None] + 1)
        if demod:
            wts = wts * torch.rsqrt((wts ** 2).sum(dim=(2, 3, 4), keepdim=True) + 1e-8)
        o = F.conv2d(xd.reshape(1, -1, Ho, Ho), wts.reshape(B * N, K, k, k), padding=k // 2, groups=B).reshape(B, N, Ho, Ho)
        if act:
            noise = F.linear(inoise[:, :Ho, :Ho, :].to(x.dtype), lw, lb).permute(0, 3, 2, 1)
            o = F.leaky_relu(o + noise, 0.2)
        return o

    nzt = inoise[..., 0].transpose(1, 2).contiguous()
    out = ops.modconv_stage(x, y, w, nzt if act else None, lin.weight if act else None, lin.bias if act else None,
                            demod=demod, upsample=up, act=act)
    dd = [t.detach().double().requires_grad_(True) for t in (x, y, w, lin.weight, lin.bias)]
    ref = reference(*dd)
    assert relmax(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-5
    go = torch.randn_like(out)
    ins = [x, y, w] + ([lin.weight, lin.bias] if act else [])
    gm = torch.autograd.grad(out, ins, go)
    gr = torch.autograd.grad(ref, dd[:len(ins)], go.double())
    for a, b in zip(gm, gr):
        assert relmax(a.cpu().numpy(), b.cpu().numpy()) <= 1e-4


def test_discriminator_with_attention_matches_oracle(gpu_device):
    """attn_layers: Residual(Rezero(ImageLinearAttention)) x 2 after the named blocks (reference :594-596), forward,
    gradient penalty (double backward through the attention) and parameter gradients against the oracle restatement
    (third-party algorithm: parity unpinned, see oracle/histogan_nets.py)."""
    from histoGAN import Discriminator
    from histoGAN.histoGAN import gradient_penalty
    from oracle import histogan_nets as N
    torch.manual_seed(4)
    D = Discriminator(32, network_capacity=2, attn_layers=[1, 3]).to(gpu_device)
    with torch.no_grad():
        for k, v in D.named_parameters():
            if k.endswith('.g'):
                v.fill_(0.6)                         # Rezero gates start at 0: open them
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    assert any(k.startswith('attn_blocks.0.1.fn.fn.to_out') for k in sd) and 'attn_blocks.1.0.fn.g' not in sd
    img = torch.rand(3, 3, 32, 32)
    x = img.clone().to(gpu_device).requires_grad_(True)
    logits, q = D(x)
    loss = torch.relu(1 + logits).mean() + gradient_penalty(x, logits)
    xc = img.clone().requires_grad_(True)
    ref_logits = N.discriminator(sd, xc, 5)
    ref_loss = torch.relu(1 + ref_logits).mean() + N.gradient_penalty(xc, ref_logits)
    assert relmax(logits.detach().cpu().numpy(), ref_logits.detach().numpy()) <= 1e-5
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    names = ['attn_blocks.0.0.fn.fn.to_q.weight', 'attn_blocks.0.1.fn.fn.to_out.bias', 'attn_blocks.2.0.fn.g',
             'attn_blocks.2.1.fn.fn.to_v.weight', 'blocks.0.net.0.weight', 'blocks.3.conv_res.weight', 'to_logit.weight']
    params = dict(D.named_parameters())
    grads = torch.autograd.grad(loss, [params[n] for n in names])
    refs = torch.autograd.grad(ref_loss, [sd[n] for n in names])
    for n, a, b in zip(names, grads, refs):
        assert relmax(a.cpu().numpy(), b.numpy()) <= 2e-4, n


def test_train_steps_with_attention(gpu_device, tmp_path):
    from histoGAN import Trainer
    tr = Trainer('attn', tmp_path / 'r', tmp_path / 'm', 32, 2, batch_size=2, hist_bin=16, hist_insz=32,
                 hist_resizing='interpolation', attn_layers=[1, 2])
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    for _ in range(2):
        tr.train(alpha=2)
    assert np.isfinite(tr.d_loss) and np.isfinite(tr.g_loss) and np.isfinite(tr.last_gp_loss)
    assert any('attn_blocks.0.0.fn.fn.to_q.weight' in k for k in tr.GAN.state_dict())


def test_vector_quantize_layers(gpu_device, tmp_path):
    """fq_layers (reference :598-600): nearest-code output, straight-through gradient and commitment loss against the
    oracle restatement; EMA codebook statistics move in training mode only; trainer steps with fq_layers."""
    from histogan_amd.nets import Discriminator, VectorQuantize
    from oracle import histogan_nets as N
    torch.manual_seed(2)
    vq = VectorQuantize(8, 32).to(gpu_device)
    x = torch.randn(3, 5, 7, 8, device=gpu_device, requires_grad=True)          # channels-last
    vq.eval()
    e0 = vq.embed.clone()
    q, loss = vq(x)
    assert torch.equal(vq.embed, e0)                                            # no update in eval mode
    xc = x.detach().cpu().requires_grad_(True)
    qr, lr = N.vector_quantize(e0.cpu(), xc)
    assert relmax(q.detach().cpu().numpy(), qr.detach().numpy()) <= 1e-6 and abs(float(loss) - float(lr)) <= 1e-6
    g, = torch.autograd.grad(q.sum() + loss, x)
    gr, = torch.autograd.grad(qr.sum() + lr, xc)
    assert relmax(g.cpu().numpy(), gr.numpy()) <= 1e-5
    codes = q.detach().reshape(-1, 8)
    assert float((codes[:, None, :] - e0.t()[None]).abs().sum(-1).min(1)[0].max()) <= 1e-5     # every output IS a code
    vq.train()
    vq(x)
    assert not torch.equal(vq.embed, e0) and float(vq.cluster_size.sum()) > 0
    D = Discriminator(32, 2, fq_layers=[2], fq_dict_size=16).to(gpu_device)
    assert {'quantize_blocks.1.fn.embed', 'quantize_blocks.1.fn.cluster_size', 'quantize_blocks.1.fn.embed_avg'} <= set(D.state_dict())
    logits, ql = D(torch.rand(2, 3, 32, 32, device=gpu_device))
    assert logits.shape == (2,) and float(ql) > 0
    from histoGAN import Trainer
    tr = Trainer('vq', tmp_path / 'r', tmp_path / 'm', 32, 2, batch_size=2, hist_bin=16, hist_insz=32,
                 hist_resizing='interpolation', fq_layers=[1, 2], fq_dict_size=16)
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    tr.train(alpha=2); tr.train(alpha=2)
    assert np.isfinite(tr.d_loss) and np.isfinite(tr.g_loss) and tr.q_loss > 0


@pytest.mark.parametrize('up', [True, False])
def test_generator_block_forward_explicit_noise(up, gpu_device):
    """GeneratorBlock.forward_ with explicit per-pixel noise tensors (the projection scripts' call, reference
    histoGAN/histoGAN.py:481-502) vs. the inoise path and vs. the reference formula written with aten ops:
    noise_k = to_noise_k(inoise[:, :H, :W]).permute(0, 3, 2, 1) (the H<->W swap of :465-466)."""
    from histogan_amd.nets import GeneratorBlock
    torch.manual_seed(4)
    B, Ci, Co, H = 2, 12, 8, 16
    blk = GeneratorBlock(32, Ci, Co, upsample=up, upsample_rgb=True).to(gpu_device)
    with torch.no_grad():
        blk.to_noise1.weight.normal_(std=0.5); blk.to_noise2.weight.normal_(std=0.5)
        blk.to_noise1.bias.normal_(std=0.1); blk.to_noise2.bias.normal_(std=0.1)
    Hin = H // 2 if up else H
    x = torch.randn(B, Ci, Hin, Hin, device=gpu_device, requires_grad=True)
    prev = torch.randn(B, 3, H, H, device=gpu_device)
    istyle = torch.randn(B, 32, device=gpu_device)
    inoise = torch.rand(B, 32, 32, 1, device=gpu_device)
    s1, s2, srgb = blk.to_style1(istyle), blk.to_style2(istyle), blk.to_rgb.to_style(istyle)
    cut = inoise[:, :H, :H, :]
    n1 = blk.to_noise1(cut).permute(0, 3, 2, 1)
    n2 = blk.to_noise2(cut).permute(0, 3, 2, 1)
    xa, rgba = blk.forward_(x, prev, s1, s2, srgb, noise1=n1, noise2=n2)
    xb, rgbb = blk.forward_(x, prev, s1, s2, srgb, inoise=inoise)
    assert relmax(xa.detach().cpu().numpy(), xb.detach().cpu().numpy()) <= 2e-6
    assert relmax(rgba.detach().cpu().numpy(), rgbb.detach().cpu().numpy()) <= 2e-6
    (ga,) = torch.autograd.grad((xa * xa).sum() + rgba.sum(), x, retain_graph=True)
    (gb,) = torch.autograd.grad((xb * xb).sum() + rgbb.sum(), x)
    assert relmax(ga.cpu().numpy(), gb.cpu().numpy()) <= 2e-5
    # the reference formula for one stage with aten ops in fp64 (Conv2DMod :420-440 on materialised per-sample weights)
    with torch.no_grad():
        xd = x.detach().double()
        if up:
            xd = F.interpolate(xd, scale_factor=2, mode='bilinear', align_corners=False)
        w = blk.conv1.weight.double()[None] * (s1.double()[:, None, :, None, None] + 1)
        w = w * torch.rsqrt((w ** 2).sum(dim=(2, 3, 4), keepdim=True) + 1e-8)
        ref = torch.cat([F.conv2d(xd[b:b + 1], w[b], padding=1) for b in range(B)])
        ref = F.leaky_relu(ref + n1.double(), 0.2)
        w2 = blk.conv2.weight.double()[None] * (s2.double()[:, None, :, None, None] + 1)
        w2 = w2 * torch.rsqrt((w2 ** 2).sum(dim=(2, 3, 4), keepdim=True) + 1e-8)
        ref = torch.cat([F.conv2d(ref[b:b + 1], w2[b], padding=1) for b in range(B)])
        ref = F.leaky_relu(ref + n2.double(), 0.2)
    assert relmax(xa.detach().cpu().numpy(), ref.cpu().numpy()) <= 1e-5
    with pytest.raises(Exception, match='No noise is given'):
        blk.forward_(x, prev, s1, s2, srgb)


@pytest.mark.parametrize('shape', [(2, 5, 4, 4), (3, 7, 16, 16), (2, 3, 2, 2), (1, 2, 64, 64), (2, 3, 5, 7), (1, 1, 3, 1)])
def test_upsample_adjoint_paths_agree(shape, gpu_device):
    """hg_modulate_bwd with upsample: the two-pixels-per-thread kernel (even widths, 16-byte-aligned rows: vector loads +
    wave shuffles) and the general one-pixel-per-thread kernel (odd widths, unaligned views) compute the same gx bit for
    bit (same fma order) and both match torch's adjoint of F.interpolate(scale_factor=2, bilinear)."""
    from histogan_amd import ops
    torch.manual_seed(8)
    B, C, H, W = shape
    x = torch.randn(B, C, H, W, device=gpu_device, requires_grad=True)
    s = torch.randn(B, C, device=gpu_device, requires_grad=True)
    go = torch.randn(B, C, 2 * H, 2 * W, device=gpu_device)
    gx, gs = torch.autograd.grad(ops.modulate(x, s, True), (x, s), go)
    ref = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) * (s + 1)[:, :, None, None]
    rx, rs = torch.autograd.grad(ref, (x, s), go)
    assert relmax(gx.cpu().numpy(), rx.cpu().numpy()) <= 1e-5 and relmax(gs.cpu().numpy(), rs.cpu().numpy()) <= 1e-5
    # the same call on a view whose storage is off by one float: the general kernel
    buf = torch.empty(go.numel() + 1, device=gpu_device)
    go2 = buf[1:].view_as(go).copy_(go)
    assert go2.data_ptr() % 16 != 0
    gx2, gs2 = torch.autograd.grad(ops.modulate(x, s, True), (x, s), go2)
    assert torch.equal(gx2, gx) and relmax(gs2.cpu().numpy(), gs.cpu().numpy()) <= 1e-6


@pytest.mark.parametrize('B,O,C,S,with_prev', [(2, 32, 3, 16, True), (3, 2048, 3, 4, True), (2, 64, 3, 128, True), (5, 40, 4, 8, False),
                                               (1, 512, 3, 2, False), (32, 32, 3, 64, True), (2, 100, 3, 6, True)])
def test_torgb_matches_fp64(B, O, C, S, with_prev, gpu_device):
    """ops.torgb (hg_torgb_fwd / hg_torgb_bwd: RGBBlock's 1x1 modulated convolution without demodulation + the running RGB
    image, histoGAN/histoGAN.py:380-390, as one stream over x per direction) against the formula in fp64: output and the
    gradients of x, style, weight and prev; deterministic; == the modulated-convolution path it replaces."""
    from histogan_amd import ops
    dev = gpu_device
    g = torch.Generator().manual_seed(B * 7 + O + S)
    x = torch.randn(B, O, S, S, generator=g).to(dev).requires_grad_(True)
    st = (0.4 * torch.randn(B, O, generator=g)).to(dev).requires_grad_(True)
    w = (torch.randn(C, O, 1, 1, generator=g) / O ** 0.5).to(dev).requires_grad_(True)
    prev = torch.randn(B, C, S, S, generator=g).to(dev).requires_grad_(True) if with_prev else None
    go = torch.randn(B, C, S, S, generator=g).to(dev)
    assert ops.torgb_supported(x, w)
    out = ops.torgb(x, st, w, prev)
    ins = [x, st, w] + ([prev] if with_prev else [])
    grads = torch.autograd.grad(out, ins, go)
    xd, sd, wd = (t.detach().double().requires_grad_(True) for t in (x, st, w))
    pd = prev.detach().double().requires_grad_(True) if with_prev else None
    ref = torch.einsum('co,bo,bohw->bchw', wd[:, :, 0, 0], sd + 1.0, xd)
    if with_prev:
        ref = ref + pd
    rgrads = torch.autograd.grad(ref, [xd, sd, wd] + ([pd] if with_prev else []), go.double())
    assert out.shape == ref.shape and relmax(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 2e-6
    for a, r in zip(grads, rgrads):
        assert a.shape == r.shape and relmax(a.cpu().numpy(), r.cpu().numpy()) <= 5e-6, (a.shape,)
    out2 = ops.torgb(x, st, w, prev)
    grads2 = torch.autograd.grad(out2, ins, go)
    assert torch.equal(out, out2) and all(torch.equal(a, b) for a, b in zip(grads, grads2))


def test_rgbblock_torgb_path_equals_modconv_path(gpu_device):
    """RGBBlock.forward_ on the fused to-RGB kernels == the same block on the modulated-convolution path (HG_TORGB off)."""
    from histogan_amd import ops
    from histoGAN import RGBBlock
    torch.manual_seed(3)
    dev = gpu_device
    blk = RGBBlock(64, 48, upsample=True).to(dev)
    x = torch.randn(3, 48, 16, 16, device=dev, requires_grad=True)
    prev = torch.randn(3, 3, 16, 16, device=dev, requires_grad=True)
    ist = torch.randn(3, 64, device=dev, requires_grad=True)
    go = torch.randn(3, 3, 32, 32, device=dev)
    params = list(blk.parameters())

    def run():
        out = blk(x, prev, ist)
        return out, torch.autograd.grad(out, [x, prev, ist] + params, go)
    o1, g1 = run()
    saved = ops.TORGB
    try:
        ops.TORGB = False
        o2, g2 = run()
    finally:
        ops.TORGB = saved
    assert relmax(o1.detach().cpu().numpy(), o2.detach().cpu().numpy()) <= 2e-6
    for a, b in zip(g1, g2):
        assert relmax(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5
