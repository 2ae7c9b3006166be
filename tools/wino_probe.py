#!/usr/bin/env python3
"""Winograd F(2x2,3x3) (include/hg_wino.h) against the direct implicit GEMM (include/hg_conv.h) at the 3x3 stride-1 layer
shapes of the C3 step: error of both vs fp64 F.conv2d, time of both (HIP events), and the fused epilogue / data-gradient /
ragged-size checks.   python tools/wino_probe.py [--batch 32] [--iters 10] [--quick]
Writes gpurun_out/wino_probe.json."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from histogan_amd import conv as C
from histogan_amd._lib import check, lib, raw_stream

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--quick', action='store_true')
ap.add_argument('--tag', default='')
args = ap.parse_args()
dev = torch.device('cuda:0')
B = args.batch
# the conv module's own entry points are the DIRECT baseline here: no automatic Winograd dispatch
C._wino_u = lambda *a, **k: None
C.wino_wgrad_supported = lambda *a: False
REPORT = {}


def p(t):
    return None if t is None else t.data_ptr()


def wino_pack(w, mode):
    Co, Ci = w.shape[:2]
    n = lib.hg_wino_packed_elems(Co, Ci, mode)
    assert n, (Co, Ci, mode)
    u = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.hg_wino_pack_weights(w.data_ptr(), u.data_ptr(), Co, Ci, mode, raw_stream(w.device)), 'hg_wino_pack_weights')
    return u


def wino_conv(x, u, N, iscale=None, oscale=None, bias=None, noise_w=None, noise_img=None, slope=0.0, addend=None):
    Bx, K, H, W = x.shape
    out = torch.empty((Bx, N, H, W), dtype=torch.float32, device=x.device)
    nb = lib.hg_wino_workspace_bytes(Bx, K, N, H, W)
    ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=x.device)
    S = noise_img.shape[-1] if noise_img is not None else 0
    check(lib.hg_wino_conv2d(x.data_ptr(), u.data_ptr(), out.data_ptr(), p(iscale), p(oscale), p(bias), p(noise_w), p(noise_img),
                             S, float(slope), p(addend), Bx, K, N, H, W, ws.data_ptr(), nb, raw_stream(x.device)), 'hg_wino_conv2d')
    return out


def wino_wgrad(x, go):
    Bx, K, H, W = x.shape
    N = go.shape[1]
    nb = lib.hg_wino_wgrad_workspace_bytes(Bx, K, N, H, W)
    assert nb, (Bx, K, N, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    gw = torch.empty((N, K, 3, 3), dtype=torch.float32, device=x.device)
    check(lib.hg_wino_wgrad(x.data_ptr(), go.data_ptr(), gw.data_ptr(), Bx, K, N, H, W, ws.data_ptr(), nb, raw_stream(x.device)),
          'hg_wino_wgrad')
    return gw


def timeit(fn, iters):
    fn()
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e-3)
    return min(ts)


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


# ---- correctness on small / ragged shapes first (cheap, catches indexing errors before the big layers) ---------------
def check_small():
    worst = 0.0
    cases = [(3, 32, 64, 8, 8), (2, 64, 128, 16, 16), (5, 40, 96, 12, 20), (1, 64, 64, 6, 10), (7, 32, 32, 8, 8),
             (2, 16, 32, 32, 32), (3, 64, 32, 24, 16), (2, 128, 192, 4, 4), (33, 64, 64, 2, 2), (2, 72, 80, 34, 30),
             (3, 72, 80, 16, 32), (5, 64, 128, 8, 4), (2, 128, 64, 64, 64)]
    for (b, K, N, H, W) in cases:
        g = torch.Generator().manual_seed(b * 131 + K + N + H)
        x = torch.randn(b, K, H, W, generator=g).to(dev)
        w = (torch.randn(N, K, 3, 3, generator=g) / (K * 9) ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        if not lib.hg_wino_packed_elems(N, K, 0):
            print('small', (b, K, N, H, W), 'not served'); continue
        ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
        out = wino_conv(x, wino_pack(w, 0), N, bias=bias)
        e = rel(out, ref)
        # data gradient: gin (b, K) <- gout (b, N): the launch's "K" is N
        go = torch.randn(b, N, H, W, generator=g).to(dev)
        e2 = None
        if lib.hg_wino_packed_elems(N, K, 1):
            refd = torch.nn.grad.conv2d_input((b, K, H, W), w.double(), go.double(), padding=1)
            outd = wino_conv(go, wino_pack(w, 1), K)
            e2 = rel(outd, refd)
        # fused epilogue
        isc = (torch.randn(b, K, generator=g) * 0.3 + 1).to(dev)
        osc = (torch.rand(b, N, generator=g) + 0.5).to(dev)
        S = max(H, W) + (max(H, W) & 1)
        nimg = torch.randn(b, S, S, generator=g).to(dev)
        nw = torch.randn(N, generator=g).to(dev)
        reff = F.conv2d(x.double() * isc.double()[:, :, None, None], w.double(), padding=1) * osc.double()[:, :, None, None] \
            + bias.double()[None, :, None, None] + nw.double()[None, :, None, None] * nimg.double()[:, None, :H, :W]
        reff = F.leaky_relu(reff, 0.2)
        outf = wino_conv(x, wino_pack(w, 0), N, iscale=isc, oscale=osc, bias=bias, noise_w=nw, noise_img=nimg, slope=0.2)
        e3 = rel(outf, reff)
        ad = torch.randn(b, N, H, W, generator=g).to(dev)
        outa = wino_conv(x, wino_pack(w, 0), N, bias=bias, addend=ad)
        e4 = rel(outa, ref + ad.double())
        e5 = None
        if lib.hg_wino_wgrad_workspace_bytes(b, K, N, H, W):
            refw = torch.nn.grad.conv2d_weight(x.double(), (N, K, 3, 3), go.double(), padding=1)
            e5 = rel(wino_wgrad(x, go), refw)
            worst = max(worst, e5)
        print(f'   wgrad {e5 if e5 is None else format(e5, ".2e")}', end='')
        print(f'small B{b} K{K} N{N} {H}x{W}: fwd {e:.2e} dgrad {e2 if e2 is None else format(e2, ".2e")} fused {e3:.2e} addend {e4:.2e}', flush=True)
        REPORT[f'small/B{b}K{K}N{N}H{H}W{W}'] = dict(fwd=e, dgrad=e2, fused=e3, addend=e4)
        worst = max(worst, e, e2 or 0, e3, e4)
    # forced K split
    return worst


def layers():
    gf = [64, 2048, 1024, 512, 256, 128, 64, 32]
    L = []
    for i in range(7):
        S = 4 * 2 ** i
        L.append((f'G{i}.conv1', B, gf[i], gf[i + 1], S))
        L.append((f'G{i}.conv2', B, gf[i + 1], gf[i + 1], S))
    df = [3, 16, 32, 64, 128, 256, 512, 1024, 2048]
    for i in range(1, 8):
        S = 256 >> i
        L.append((f'D{i}.c1', 2 * B, df[i], df[i + 1], S))
        L.append((f'D{i}.c2', 2 * B, df[i + 1], df[i + 1], S))
    return L


def main():
    worst = check_small()
    print('small-shape worst error %.2e' % worst, flush=True)
    rows = []
    tot = dict(d=0.0, w=0.0, dd=0.0, wd=0.0)
    print(f'{"layer":10} {"B":>3} {"K":>5} {"N":>5} {"S":>4} | direct ms    TF   err | wino ms  eff.TF   err  speedup | dgrad: direct ms  wino ms  err  speedup')
    for tag, b, K, N, S in layers():
        if args.quick and S not in (8, 64, 256):
            continue
        g = torch.Generator().manual_seed(K * 13 + N * 3 + S)
        x = torch.randn(b, K, S, S, generator=g).to(dev)
        w = (torch.randn(N, K, 3, 3, generator=g) / (K * 9) ** 0.5).to(dev)
        go = torch.randn(b, N, S, S, generator=g).to(dev)
        flops = 2.0 * b * S * S * K * N * 9
        wf, wd = C.pack_weights(w, C.PACK_FWD), C.pack_weights(w, C.PACK_DGRAD)
        td = timeit(lambda: C.conv_fwd_packed(x, wf, N, 3), args.iters)
        tdd = timeit(lambda: C.conv_dgrad_packed(go, wd, K, S, S, 3), args.iters)
        row = dict(layer=tag, B=b, K=K, N=N, S=S, direct_ms=td * 1e3, direct_dgrad_ms=tdd * 1e3)
        line = f'{tag:10} {b:3d} {K:5d} {N:5d} {S:4d} | {td*1e3:8.3f} {flops/td/1e12:6.1f}'
        nref = min(b, 4)   # fp64 reference on a slice of the batch (error is per sample)
        ref = F.conv2d(x[:nref].double(), w.double(), padding=1)
        ed = rel(C.conv_fwd_packed(x, wf, N, 3)[:nref], ref)
        line += f' {ed:.1e} |'
        if lib.hg_wino_packed_elems(N, K, 0) and lib.hg_wino_workspace_bytes(b, K, N, S, S) < 2 ** 33:
            u = wino_pack(w, 0)
            ow = wino_conv(x, u, N)
            ew = rel(ow[:nref], ref)
            tw = timeit(lambda: wino_conv(x, u, N), args.iters)
            row.update(wino_ms=tw * 1e3, wino_err=ew, direct_err=ed, speedup=td / tw,
                       supported=int(lib.hg_wino_supported(b, K, N, S, S)))
            line += f' {tw*1e3:8.3f} {flops/tw/1e12:6.1f} {ew:.1e} {td/tw:6.2f}x |'
            tot['d'] += td; tot['w'] += tw
        else:
            line += ' (not served) |'
        if lib.hg_wino_packed_elems(N, K, 1):
            ud = wino_pack(w, 1)
            refd = torch.nn.grad.conv2d_input((nref, K, S, S), w.double(), go[:nref].double(), padding=1)
            owd = wino_conv(go, ud, K)
            ewd = rel(owd[:nref], refd)
            twd = timeit(lambda: wino_conv(go, ud, K), args.iters)
            row.update(wino_dgrad_ms=twd * 1e3, wino_dgrad_err=ewd, dgrad_speedup=tdd / twd)
            line += f' {tdd*1e3:8.3f} {twd*1e3:8.3f} {ewd:.1e} {tdd/twd:6.2f}x'
            tot['dd'] += tdd; tot['wd'] += twd
        if lib.hg_wino_wgrad_workspace_bytes(b, K, N, S, S) and K >= 32 and N >= 32:
            tw0 = timeit(lambda: C.conv_wgrad(x, go, 3), args.iters)
            gwd = C.conv_wgrad(x, go, 3)
            tw1 = timeit(lambda: wino_wgrad(x, go), args.iters)
            gww = wino_wgrad(x, go)
            # fp64 reference of a slice of the weight (all pixels, all samples): the first 8 output channels
            nn = min(N, 8)
            refw = torch.nn.grad.conv2d_weight(x.double(), (nn, K, 3, 3), go[:, :nn].double(), padding=1)
            ewd, eww = rel(gwd[:nn], refw), rel(gww[:nn], refw)
            row.update(direct_wgrad_ms=tw0 * 1e3, wino_wgrad_ms=tw1 * 1e3, direct_wgrad_err=ewd, wino_wgrad_err=eww,
                       wgrad_speedup=tw0 / tw1, wgrad_supported=int(lib.hg_wino_wgrad_supported(b, K, N, S, S)))
            line += f' | wgrad {tw0*1e3:8.3f} {ewd:.1e} -> {tw1*1e3:8.3f} {eww:.1e} {tw0/tw1:5.2f}x'
            tot['gd'] = tot.get('gd', 0.0) + tw0; tot['gw'] = tot.get('gw', 0.0) + tw1
        print(line, flush=True)
        rows.append(row)
        del x, w, go
    print('sum over served layers: fwd direct %.2f ms -> wino %.2f ms;  dgrad direct %.2f ms -> wino %.2f ms;  wgrad %.2f -> %.2f ms' % (
        tot['d'] * 1e3, tot['w'] * 1e3, tot['dd'] * 1e3, tot['wd'] * 1e3, tot.get('gd', 0) * 1e3, tot.get('gw', 0) * 1e3))
    REPORT['layers'] = rows
    REPORT['totals_ms'] = {k: v * 1e3 for k, v in tot.items()}
    os.makedirs('gpurun_out', exist_ok=True)
    with open(f'gpurun_out/wino_probe{args.tag}.json', 'w') as f:
        json.dump(REPORT, f, indent=1)


if __name__ == '__main__':
    main()
