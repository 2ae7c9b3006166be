"""Functional torch restatement of the reference's networks and losses.  TEST INFRASTRUCTURE ONLY.

Every function takes a flat state dict with the reference's parameter names
(`blocks.3.conv1.weight`, ...) so reference checkpoints / golden weights plug in directly.
Pinned against golden vectors produced by the unmodified reference classes
(tests/golden/make_golden_nets.py -> nets_small.npz; test: tests/test_oracle_nets_golden.py).

Follows (cites relative to the reference root):
* conv2d_mod ............ histoGAN/histoGAN.py:420-440  (per-sample weights, ONE grouped conv)
* rgb_block ............. histoGAN/histoGAN.py:380-390
* generator_block ....... histoGAN/histoGAN.py:461-479  (noise permute (0,3,2,1): H<->W swapped)
* generator ............. histoGAN/histoGAN.py:558-568
* discriminator[_block] . histoGAN/histoGAN.py:520-526, 613-631 (no VQ layers; attention: see image_linear_attention)
* vectorizer ............ histoGAN/histoGAN.py:335-365
* gradient_penalty ...... histoGAN/histoGAN.py:156-163
* diffgrad_step ......... torch_optimizer.DiffGrad (third-party, source NOT in the reference tree,
                          version unpinned: restated from the upstream algorithm -- PARITY UNPINNED)
"""
import math

import torch
import torch.nn.functional as F

EPS = 1e-8  # histoGAN/histoGAN.py:53


def lrelu(x):
    return F.leaky_relu(x, 0.2)


def conv2d_mod(x, y, weight, demod=True):
    b, c, h, w = x.shape
    co, ci, k, _ = weight.shape
    wts = weight[None] * (y[:, None, :, None, None] + 1)
    if demod:
        wts = wts * torch.rsqrt((wts ** 2).sum(dim=(2, 3, 4), keepdim=True) + EPS)
    out = F.conv2d(x.reshape(1, -1, h, w), wts.reshape(b * co, ci, k, k), padding=(k - 1) // 2, groups=b)
    return out.reshape(-1, co, h, w)


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


def _lin(sd, name, x):
    return F.linear(x, sd[name + '.weight'], sd[name + '.bias'])


def rgb_block(sd, pre, x, prev_rgb, istyle, upsample):
    x = conv2d_mod(x, _lin(sd, pre + 'to_style', istyle), sd[pre + 'conv.weight'], demod=False)
    if prev_rgb is not None:
        x = x + prev_rgb
    return _up2(x) if upsample else x


def generator_block(sd, pre, x, prev_rgb, istyle, inoise, upsample, upsample_rgb):
    if upsample:
        x = _up2(x)
    inoise = inoise[:, :x.shape[2], :x.shape[3], :]
    n1 = _lin(sd, pre + 'to_noise1', inoise).permute(0, 3, 2, 1)
    n2 = _lin(sd, pre + 'to_noise2', inoise).permute(0, 3, 2, 1)
    x = lrelu(conv2d_mod(x, _lin(sd, pre + 'to_style1', istyle), sd[pre + 'conv1.weight']) + n1)
    x = lrelu(conv2d_mod(x, _lin(sd, pre + 'to_style2', istyle), sd[pre + 'conv2.weight']) + n2)
    return x, rgb_block(sd, pre + 'to_rgb.', x, prev_rgb, istyle, upsample_rgb)


def generator(sd, styles, hists, noise, num_layers):
    """styles (B, L-2, latent), hists (B, 2, latent), noise (B, S, S, 1) -> rgb (B, 3, S, S)."""
    x = sd['initial_block'].expand(styles.shape[0], -1, -1, -1)
    sty = torch.cat((styles.transpose(0, 1), hists.transpose(0, 1)), dim=0)
    rgb = None
    for i in range(num_layers):
        x, rgb = generator_block(sd, f'blocks.{i}.', x, rgb, sty[i], noise,
                                 upsample=i != 0, upsample_rgb=i != num_layers - 1)
    return rgb


def image_linear_attention(sd, pre, x, key_dim=64, heads=8):
    """linear_attention_transformer.images.ImageLinearAttention (third-party, not in the reference tree, version
    unpinned: restated from the published algorithm -- PARITY UNPINNED); defaults as the reference calls it
    (histoGAN/histoGAN.py:594-596: 1x1 projections, key_dim = value_dim = 64, heads = 8, norm_queries)."""
    b, c, h, w = x.shape
    q, k, v = (F.conv2d(x, sd[f'{pre}to_{n}.weight'], sd[f'{pre}to_{n}.bias']).reshape(b, heads, -1, h * w) for n in 'qkv')
    q, k = q * key_dim ** -0.25, k * key_dim ** -0.25
    k = k.softmax(dim=-1)
    q = q.softmax(dim=-2)
    ctx = torch.einsum('bhdn,bhen->bhde', k, v)
    out = torch.einsum('bhdn,bhde->bhen', q, ctx).reshape(b, -1, h, w)
    return F.conv2d(out, sd[pre + 'to_out.weight'], sd[pre + 'to_out.bias'])


def vector_quantize(embed, x, commitment=1.0):
    """vector_quantize_pytorch.VectorQuantize in eval mode (no EMA update) on a channels-last tensor: nearest code,
    straight-through output, commitment loss (third-party, unpinned: restated -- PARITY UNPINNED)."""
    flat = x.reshape(-1, embed.shape[0])
    dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ embed + embed.pow(2).sum(0, keepdim=True)
    ind = (-dist).max(1)[1]
    q = F.embedding(ind.view(*x.shape[:-1]), embed.t())
    return x + (q - x).detach(), F.mse_loss(q.detach(), x) * commitment


def discriminator(sd, x, num_blocks):
    for i in range(num_blocks):
        p = f'blocks.{i}.'
        res = F.conv2d(x, sd[p + 'conv_res.weight'], sd[p + 'conv_res.bias'])
        x = lrelu(F.conv2d(x, sd[p + 'net.0.weight'], sd[p + 'net.0.bias'], padding=1))
        x = lrelu(F.conv2d(x, sd[p + 'net.2.weight'], sd[p + 'net.2.bias'], padding=1))
        x = x + res
        if p + 'downsample.weight' in sd:
            x = F.conv2d(x, sd[p + 'downsample.weight'], sd[p + 'downsample.bias'], padding=1, stride=2)
        for j in range(2):     # Residual(Rezero(ImageLinearAttention)) x 2 on the layers named in attn_layers (:594-596)
            a = f'attn_blocks.{i}.{j}.fn.'
            if a + 'g' in sd:
                x = image_linear_attention(sd, a + 'fn.', x) * sd[a + 'g'] + x
    x = x.reshape(x.shape[0], -1)
    return _lin(sd, 'to_logit', x).squeeze()


def vectorizer(sd, x, seq='net'):
    """StyleVectorizer (seq='net') / HistVectorizer (seq='fcs', input flattened)."""
    x = x.reshape(x.shape[0], -1)
    i = 0
    while f'{seq}.{i}.weight' in sd:
        x = lrelu(_lin(sd, f'{seq}.{i}', x))
        i += 2
    return x


def gradient_penalty(images, output, weight=10):
    g, = torch.autograd.grad(output, images, torch.ones_like(output), create_graph=True)
    g = g.reshape(images.shape[0], -1)
    return weight * ((g.norm(2, dim=1) - 1) ** 2).mean()


def styles_def_to_tensor(styles_def):
    return torch.cat([t[:, None, :].expand(-1, n, -1) for t, n in styles_def], dim=1)


def diffgrad_step(p, grad, state, lr, betas=(0.5, 0.9), eps=1e-8):
    """One DiffGrad update of tensor p (in place).  state: dict(step, exp_avg, exp_avg_sq, previous_grad)."""
    b1, b2 = betas
    state['step'] += 1
    state['exp_avg'].mul_(b1).add_(grad, alpha=1 - b1)
    state['exp_avg_sq'].mul_(b2).addcmul_(grad, grad, value=1 - b2)
    denom = state['exp_avg_sq'].sqrt().add_(eps)
    bc1, bc2 = 1 - b1 ** state['step'], 1 - b2 ** state['step']
    dfc = 1.0 / (1.0 + torch.exp(-torch.abs(state['previous_grad'] - grad)))
    state['previous_grad'] = grad.clone()
    p.addcdiv_(state['exp_avg'] * dfc, denom, value=-lr * math.sqrt(bc2) / bc1)
