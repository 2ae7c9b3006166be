"""Test helper (NOT a test module): one HistoGAN optimisation step evaluated with oracle/ -- the functional
restatement of the reference's networks, histogram block, Hellinger loss and DiffGrad -- in any dtype on any device.

Follows the reference step line by line (histoGAN/histoGAN.py:853-1020): D phase (generator under no_grad, hinge
loss, gradient penalty on gradient-penalty steps), DiffGrad on D, G phase against the UPDATED discriminator
(adversarial + alpha * Hellinger on relu(G(z)), image-space path-length term on path-length steps), DiffGrad on
G / S / H.  dtype float64 gives the "truth" the 2x criterion of SURVEY.md section 8c measures distances to; float32
on the GPU is the reference's own numerics on this machine (aten / MIOpen / rocBLAS).
"""
import torch
import torch.nn.functional as F


class ReplayRng:
    """Feeds Trainer.train the tensors the oracle step uses (same draw order as the reference, :166-189)."""

    mode = 'replay'

    def __init__(self, device, B, L, LAT, S, seed, tt=1, dtype=torch.float32):
        g = torch.Generator().manual_seed(seed)
        self.dev, self.dtype = device, dtype
        self.z = [torch.randn(B, LAT, generator=g) for _ in range(4)]
        self.img_noise = [torch.rand(B, S, S, 1, generator=g) for _ in range(2)]
        self.pl = torch.randn(B, L - 2, LAT, generator=g)
        self.zi = self.ni = 0
        self.tt = tt

    def _t(self, x):
        return x.to(self.dev).to(self.dtype)

    def noise(self, n, d):
        z = self.z[self.zi]; self.zi += 1
        return self._t(z)

    def noise_list(self, n, layers, d):
        return [(self.noise(n, d), layers)]

    def mixed_list(self, n, layers, d):
        return self.noise_list(n, self.tt, d) + self.noise_list(n, layers - self.tt, d)

    def image_noise(self, n, s):
        x = self.img_noise[self.ni]; self.ni += 1
        return self._t(x)

    def randn_like(self, t):
        return self._t(self.pl)


def oracle_train_step(sd0, batches, rng, L, HB, alpha, lr, gp, pl, pl_mean=0, hist_kw=None, optimizer=True, d_override=None,
                      split_d=False):
    """sd0: GAN.state_dict() before the step (any device / dtype; moved to rng's).  batches: [D-phase batch, G-phase
    batch] of {'images', 'histograms'}.  gp / pl: gradient-penalty / path-length step.  Returns a dict with the loss
    values, the gradients {('D', name): g, ('G', name): g, ...} and (optimizer=True) the updated parameters.
    d_override: {name: tensor} discriminator parameters to use in the G phase INSTEAD of the oracle's own updated ones.
    The first DiffGrad step moves every parameter by ~lr * sign(g): where a gradient is rounding noise its sign -- and with
    it the parameter after the step -- is ill-conditioned, and at network_capacity 16 the 9e7 such +-lr choices shift the
    (un-normalised, |logit| >> 1) discriminator output visibly.  Handing the G phase the discriminator the product path
    actually used separates the two questions: the D update is checked on its own (masked parameter deltas), the G phase
    against the same discriminator."""
    from oracle import histogan_nets as N
    from oracle import rgbuv_hist as OH
    dev, dt = rng.dev, rng.dtype
    hist_kw = dict(hist_kw or {})
    cvt = lambda t: t.detach().to(dev).to(dt)
    sub = lambda p: {k[len(p) + 1:]: cvt(v).clone().requires_grad_(True) for k, v in sd0.items()
                     if k.startswith(p + '.') and v.dtype.is_floating_point}
    sG, sD, sS, sH = sub('G'), sub('D'), sub('S'), sub('H')
    nblk = L + 1
    B = batches[0]['images'].shape[0]
    LAT = sG['blocks.0.to_style1.weight'].shape[1]
    S_ = batches[0]['images'].shape[-1]

    def w_hw(style, hist):
        w = [(N.vectorizer(sS, z, 'net'), n) for z, n in style]
        hw = N.vectorizer(sH, hist, 'fcs')[:, None, :]
        return N.styles_def_to_tensor(w), torch.cat((hw, hw), 1)

    def diffgrad(params, grads):
        for k, gr in zip(params, grads):
            st = dict(step=0, exp_avg=torch.zeros_like(gr), exp_avg_sq=torch.zeros_like(gr),
                      previous_grad=torch.zeros_like(gr))
            with torch.no_grad():
                N.diffgrad_step(params[k], gr, st, lr=lr, betas=(0.5, 0.9))

    out = {}
    # ---- D phase (:889-932)
    style = rng.mixed_list(B, L - 2, LAT); noise = rng.image_noise(B, S_)
    img = cvt(batches[0]['images']).clone().requires_grad_(True)
    with torch.no_grad():
        w, hw = w_hw(style, cvt(batches[0]['histograms']))
        fake = N.generator(sG, w, hw, noise, L)
    fake_out = N.discriminator(sD, fake, nblk)
    real_out = N.discriminator(sD, img, nblk)
    div = (F.relu(1 + real_out) + F.relu(1 - fake_out)).mean()
    d_loss = div
    if gp:
        gpv = N.gradient_penalty(img, real_out)
        d_loss = d_loss + gpv
        out['gp'] = float(gpv)
    dk = list(sD.keys())
    if split_d:
        # The hinge's two halves separately: d_scale = |gradient of the real half (+ penalty)| + |gradient of the fake half|.
        # With every sample inside the hinge the logit gradients are +1/2B (real) and -1/2B (fake), and a bias gradient --
        # a plain sum of the back-propagated signal over batch and pixels -- is the difference of two nearly equal halves:
        # its fp32 error is a multiple of eps * d_scale, not of eps * |gradient|.  Tests divide bias-gradient errors by
        # max(d_scale) (the un-cancelled magnitude) instead of max |gradient|.
        real_part = F.relu(1 + real_out).mean() + (gpv if gp else 0.0)
        g_r = torch.autograd.grad(real_part, [sD[k] for k in dk], retain_graph=True)
        g_f = torch.autograd.grad(F.relu(1 - fake_out).mean(), [sD[k] for k in dk], retain_graph=True)
        out['d_scale'] = {k: float((a.abs() + b.abs()).max()) for k, a, b in zip(dk, g_r, g_f)}
        del g_r, g_f
    dgr = torch.autograd.grad(d_loss, [sD[k] for k in dk])
    out['d_loss'] = float(div.detach())
    grads = {('D', k): g.detach() for k, g in zip(dk, dgr)}
    if optimizer:
        diffgrad({k: sD[k] for k in dk}, dgr)
    out['params_d'] = {k: v.detach().clone() for k, v in sD.items()}
    if d_override is not None:
        sD = {k: cvt(d_override[k]).clone().requires_grad_(True) for k in sD}
    del fake_out, real_out, div, d_loss, dgr
    if gp:
        del gpv
    # ---- G phase (:934-989), against the UPDATED discriminator
    style = rng.mixed_list(B, L - 2, LAT); noise = rng.image_noise(B, S_)
    tgt = cvt(batches[1]['histograms'])
    w, hw = w_hw(style, tgt)
    gen_img = N.generator(sG, w, hw, noise, L)
    fo = N.discriminator(sD, gen_img, nblk)
    gh = OH.rgbuv_hist(F.relu(gen_img), h=HB, truth=dt == torch.float64, **hist_kw)
    h_loss = OH.hellinger_loss(tgt, gh.to(dt), alpha)
    g_loss = fo.mean() + h_loss
    if pl:
        std = 0.1 / (w.std(dim=0, keepdim=True) + 1e-8)
        w2 = w + rng.randn_like(w) / (std + 1e-8)
        pll = ((N.generator(sG, w2, hw, noise, L) - gen_img) ** 2).mean(dim=(1, 2, 3))
        out['pl'] = float(pll.mean())
        g_loss = g_loss + ((pll - pl_mean) ** 2).mean()
    groups = [('G', sG), ('S', sS), ('H', sH)]
    keys = [(p, k) for p, s in groups for k in s]
    ggr = torch.autograd.grad(g_loss, [dict(groups)[p][k] for p, k in keys])
    out['g_loss'], out['h_loss'] = float(fo.mean()), float(h_loss)
    out['gen_img'] = gen_img.detach()
    grads.update({pk: g.detach() for pk, g in zip(keys, ggr)})
    if optimizer:
        for p, s in groups:
            ks = [k for pp, k in keys if pp == p]
            diffgrad({k: s[k] for k in ks}, [grads[(p, k)] for k in ks])
    out['grads'] = grads
    out['params'] = {(p, k): v.detach() for p, s in [('D', out['params_d'])] + groups for k, v in s.items()}
    return out


def lrelu_margin(sd_d, images, nblk):
    """Smallest |pre-activation| of any LeakyReLU of the discriminator on `images`, relative to its layer's largest, evaluated
    in fp64.  A pre-activation within fp32 rounding of zero (~1e-7 relative) gets the other LeakyReLU slope in ANY fp32
    evaluation -- ours, aten's -- than in fp64; the gradient penalty's second-order terms then differ by that one pixel's
    whole contribution (at B = 2 ~1e-2 of a weight gradient; measured: element [1, 3, 19, 25] of blocks.3.net.2, |pre| / max
    = 1.3e-9, tools/debug_gp4.py).  Tests that hold discriminator gradients of a gradient-penalty step to 1e-4 pick input
    data whose margin is well above fp32 resolution."""
    sd = {k: v.detach().double() for k, v in sd_d.items()}
    x = images.detach().double()
    m = 1.0
    # (runs where the images live: on the GPU in the C3 test, ~0.2 s per candidate batch)
    for i in range(nblk):
        p = f'blocks.{i}.'
        pre = F.conv2d(x, sd[p + 'net.0.weight'], sd[p + 'net.0.bias'], padding=1)
        m = min(m, float(pre.abs().min() / pre.abs().max()))
        h = F.leaky_relu(pre, 0.2)
        pre = F.conv2d(h, sd[p + 'net.2.weight'], sd[p + 'net.2.bias'], padding=1)
        m = min(m, float(pre.abs().min() / pre.abs().max()))
        x = F.leaky_relu(pre, 0.2) + F.conv2d(x, sd[p + 'conv_res.weight'], sd[p + 'conv_res.bias'])
        if p + 'downsample.weight' in sd:
            x = F.conv2d(x, sd[p + 'downsample.weight'], sd[p + 'downsample.bias'], padding=1, stride=2)
        for j in range(2):          # Residual(Rezero(ImageLinearAttention)) x 2 on attention layers (no LeakyReLU inside)
            a = f'attn_blocks.{i}.{j}.fn.'
            if a + 'g' in sd:
                from oracle import histogan_nets as N
                x = N.image_linear_attention(sd, a + 'fn.', x) * sd[a + 'g'] + x
    return m


class LreluMargin:
    """Context manager: while active, every LeakyReLU of the oracle networks (oracle.histogan_nets.lrelu) on a feature map
    records min |pre-activation| / max |pre-activation|; `.value` is the smallest seen.  Run around an fp64 oracle pass to
    learn how far the closest pre-activation of that input is from zero (see lrelu_margin)."""

    def __init__(self):
        self.value = 1.0

    def __enter__(self):
        from oracle import histogan_nets as N
        self._N, self._orig = N, N.lrelu

        def hooked(x):
            if x.dim() == 4:
                a = x.detach().abs()
                self.value = min(self.value, float(a.min() / a.max()))
            return self._orig(x)

        N.lrelu = hooked
        return self

    def __exit__(self, *exc):
        self._N.lrelu = self._orig
        return False


class LreluMasks:
    """Same-branch comparison for LeakyReLU networks.  An fp32 evaluation and the fp64 oracle disagree on the slope of the
    few pre-activations that lie within fp32 rounding of zero (at 256^2 / capacity 16 / B = 2 the generator has 11 M of
    them, the closest ~1e-8 of its layer's maximum, so most input draws have a handful), and ONE such pixel moves small
    gradient tensors (to_noise, the 4x4 ... 16x16 weights) by 1e-3 -- for aten's fp32 as for ours (tools/skinny_check.py:
    5 of 8 draws).  That is a property of fp32, not an arithmetic error, and it makes a max-norm gradient bar a coin toss.
    So the oracle is evaluated on the SAME branches: `masks` (out > 0 of every feature-map LeakyReLU of the run under
    test, in call order) replace the oracle's own sign decisions; what is left is arithmetic error.  `flips` counts the
    disagreeing elements and `flip_margin` the largest |pre| / max|pre| among them (they must all be rounding-sized).
    A `None` entry (and every call past the end of the list) keeps the oracle's own decisions for that call."""

    def __init__(self, masks):
        self.masks, self.k, self.flips, self.flip_margin, self.total = list(masks), 0, 0, 0.0, 0

    def __enter__(self):
        from oracle import histogan_nets as N
        self._N, self._orig = N, N.lrelu

        def hooked(x):
            if x.dim() != 4:
                return self._orig(x)
            m = self.masks[self.k] if self.k < len(self.masks) else None
            self.k += 1
            if m is None:                # no recorded branch decisions for this call: the oracle's own
                return self._orig(x)
            assert m.shape == x.shape, (m.shape, x.shape)
            own = x.detach() > 0
            diff = own != m
            n = int(diff.sum())
            self.total += x.numel()
            if n:
                self.flips += n
                self.flip_margin = max(self.flip_margin, float(x.detach().abs()[diff].max() / x.detach().abs().max()))
            return torch.where(m, x, 0.2 * x)

        N.lrelu = hooked
        return self

    def __exit__(self, *exc):
        self._N.lrelu = self._orig
        return False
