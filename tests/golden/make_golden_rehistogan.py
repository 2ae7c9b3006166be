#!/usr/bin/env python3
"""Golden vectors for the ReHistoGAN train-step pieces, from the UNMODIFIED
reference classes of ReHistoGAN/rehistoGAN.py.

    python tests/golden/make_golden_rehistogan.py      # writes tests/golden/rehistogan_small.npz

The reference module hard-imports torch_optimizer, torchvision, cv2 (through utils.pyramid_upsampling),
vector_quantize_pytorch, linear_attention_transformer and asserts CUDA at import; those names are stubbed in
sys.modules and torch.cuda.is_available / current_device are patched, so the reference's own EncoderBlock /
DecoderBlock / RecoloringEncoderDecoder / RecoloringGAN / reconstruction_loss / get_gaussian_kernel run on CPU.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ('torch_optimizer', 'torchvision', 'torchvision.transforms', 'vector_quantize_pytorch',
                 'linear_attention_transformer', 'retry', 'retry.api', 'cv2'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['torch_optimizer'].DiffGrad = object
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['vector_quantize_pytorch'].VectorQuantize = object
    sys.modules['linear_attention_transformer'].ImageLinearAttention = object
    sys.path.insert(0, REF)
    real = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        import ReHistoGAN.rehistoGAN as R
    finally:
        torch.cuda.is_available = real
        sys.path.pop(0)
    return R


def main():
    R = import_reference()
    torch.cuda.current_device = lambda: 'cpu'
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.manual_seed(0)
    S_, CAP, LAT, HB, B = 64, 2, 32, 16, 2
    out = {}

    def put(prefix, sd):
        for k, v in sd.items():
            out[f'{prefix}/{k}'] = v.detach().numpy()

    img = torch.rand(B, 3, S_, S_)
    hist = torch.rand(B, 3, HB, HB); hist = hist / hist.sum(dim=(1, 2, 3), keepdim=True)
    noise = torch.rand(B, S_, S_, 1)
    out.update(img=img.numpy(), hist=hist.numpy(), noise=noise.numpy())
    HV = R.HistVectorizer(HB, LAT, 3)
    put('H', HV.state_dict())
    hw = HV(hist).detach()
    out['hw'] = hw.numpy()

    # three encoder-decoder variants (default; skip connections; skip + internal histogram) x the recolouring head, forward + a few gradients
    for tag, (skip, internal) in dict(plain=(False, False), skip=(True, False), skipint=(True, True)).items():
        ED = R.RecoloringEncoderDecoder(S_, network_capacity=CAP, hist=HB, latent_dim=LAT, style_depth=3,
                                        skip_conn_to_GAN=skip, internal_hist=internal)
        G = R.RecoloringGAN(S_, LAT, CAP)
        for blk in G.blocks:
            for lin in (blk.to_noise1, blk.to_noise2):
                torch.nn.init.normal_(lin.weight, std=0.5)
                torch.nn.init.normal_(lin.bias, std=0.1)
        put(f'{tag}/ED', ED.state_dict()); put(f'{tag}/G', G.state_dict())
        x = img.clone().requires_grad_(True)
        h_in = hw if internal else hist
        res = ED(x, h_in)
        if skip:
            lat, rgb, p1, p2 = res
            gen = G(lat, rgb, hw, noise, p1, p2)
        else:
            lat, rgb = res
            gen = G(lat, rgb, hw, noise)
        go = torch.randn_like(gen)
        edp, gp = dict(ED.named_parameters()), dict(G.named_parameters())
        names = ['mapping.weight', 'encoder_blocks.0.net.0.weight', 'encoder_blocks.1.net.3.bias',
                 'encoder_blocks.2.downsample.weight', 'decoder_blocks.0.block2.0.weight',
                 'decoder_blocks.1.conv_out_rgb.weight', 'decoder_mapping.bias']
        if skip:
            names += ['conv_latent_1.weight', 'to_latent_2.weight']
        if internal:
            names += ['decoder_blocks.0.conv_latent.weight', 'decoder_blocks.1.to_latent.bias']
        gnames = ['blocks.0.conv1.weight', 'blocks.1.to_rgb.conv.weight', 'blocks.1.to_noise1.weight']
        # decoder_blocks.*.conv_out_rgb feed only the rgb the head discards: their gradient is None
        used = [n for n in names if 'conv_out_rgb' not in n]
        grads = torch.autograd.grad(gen, [x] + [edp[n] for n in used] + [gp[n] for n in gnames], go)
        out.update({f'{tag}/latent': lat.detach().numpy(), f'{tag}/rgb': rgb.detach().numpy(),
                    f'{tag}/gen': gen.detach().numpy(), f'{tag}/go': go.numpy(), f'{tag}/gx': grads[0].numpy()})
        if skip:
            out.update({f'{tag}/p1': p1.detach().numpy(), f'{tag}/p2': p2.detach().numpy()})
        for n, g in zip(used, grads[1:1 + len(used)]):
            out[f'{tag}/ed_grad/{n}'] = g.numpy()
        for n, g in zip(gnames, grads[1 + len(used):]):
            out[f'{tag}/g_grad/{n}'] = g.numpy()

    # losses
    a = torch.rand(B, 3, 40, 40); b = (a + 0.1 * torch.randn_like(a)).requires_grad_(True)
    out.update(loss_a=a.numpy(), loss_b=b.detach().numpy())
    for kind, tag in (('L1', 'l1'), ('1st gradient', 'sobel'), ('2nd gradient', 'lap')):
        f = R.reconstruction_loss(kind)
        val = f.compute_loss(a, b)
        g, = torch.autograd.grad(val, b)
        out[f'rec_{tag}'] = np.float64(val.item()); out[f'rec_{tag}_grad'] = g.numpy()
    gk = R.get_gaussian_kernel(kernel_size=15, sigma=5, channels=3)
    out['gauss_k'] = gk.weight.detach().numpy()
    blur = R.gaussian_op(a, kernel=gk)
    out['gauss_out'] = blur.detach().numpy()
    h2 = torch.rand(B, 3, HB, HB); h2 = h2 / h2.sum(dim=(1, 2, 3), keepdim=True)
    out['hist2'] = h2.numpy()
    beta = 1.5
    var = -1 * (beta / 10) * torch.sum(torch.abs(hist - h2)) * torch.mean(torch.abs(
        torch.std(torch.std(R.gaussian_op(a, kernel=gk), dim=2), dim=2) -
        torch.std(torch.std(R.gaussian_op(b, kernel=gk), dim=2), dim=2)))      # :1025-1029 verbatim terms
    g, = torch.autograd.grad(var, b)
    out['var_loss'] = np.float64(var.item()); out['var_grad'] = g.numpy()
    hl = 32 * R.SCALE * (torch.sqrt(torch.sum(torch.pow(torch.sqrt(hist) - torch.sqrt(h2), 2)))) / hist.shape[0]
    out['hist_loss'] = np.float64(hl.item())
    out['meta'] = np.array([S_, CAP, LAT, HB, B])
    np.savez_compressed(os.path.join(HERE, 'rehistogan_small.npz'), **out)
    print('wrote rehistogan_small.npz with', len(out), 'arrays,',
          os.path.getsize(os.path.join(HERE, 'rehistogan_small.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
