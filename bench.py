#!/usr/bin/env python3
"""bench.py -- hot-path benchmark (contract: one JSON line on rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train|hist]

`--gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset) starts N ranks itself, one per GPU (`launch_ranks`);
under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the N ranks.

Workload `train` (default; BASELINE.json configs[2], and configs[3] under --gpus 8): the full HistoGAN
G+D train step (histogan_amd/trainer.py == reference Trainer.train, histoGAN/histoGAN.py:853-1020) at
256^2, network_capacity 16, h=64, batch 32 PER GPU on resident synthetic data; the K timed steps
follow W warm-up steps, so with the defaults (W=4, K=32) they contain 8 gradient-penalty steps and
1 path-length step -- the reference's steady-state mix.  evaluate()/save() are excluded.  Multi-GPU:
one process per GPU, RCCL all-reduce of the flat gradient buffers (weak scaling).

Workload `hist` (BASELINE.json configs[1]): RGB-uv histogram forward + Hellinger loss + backward on a
resident batch 32x3x256x256, h=64, inverse-quadratic sigma=0.02, insz=256 (N = 65 536 pixels/image).

`roofline` is the dominant hand-written kernel of the workload, timed live with HIP events around its C-ABI
call: `train` -> k_conv (the fp32-MFMA implicit-GEMM convolution; >50 % of the step), at the generator layer
256->128 channels, 64x64, batch 32 (the 128ch x 128px tile instantiation), with the weight-gradient kernel,
the per-pass aggregates over all generator 3x3 layers and the histogram kernels as sub-objects;
`hist` -> k_hist_bwd.  `cpu_baseline` = the oracle (CPU restatement of the reference) on the host cores.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: what RCCL / cross-process device memory need on these hosts

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = 'train images/sec (G+D step, 256², h=64) at 1/2/4/8 MI355X; hist-kernel HBM GB/s'
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA (v_mfma_f32_32x32x2_f32)
HBM_PEAK_GBPS = 8000.0


def hist_workload(args, dev, rank, world):
    from histogan_amd.hist import HistConfig, hellinger_loss, rgbuv_hist
    B, S, h = args.batch, args.size, args.bins
    insz = args.hist_insz or S
    cfg = HistConfig(h=h, insz=insz, method='inverse-quadratic', sigma=0.02)
    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    x = torch.rand(B, 3, S, S, generator=g).to(dev).requires_grad_(True)
    with torch.no_grad():
        target = rgbuv_hist(torch.rand(B, 3, S, S, generator=g).to(dev), cfg)

    def step():
        x.grad = None
        hist = rgbuv_hist(x, cfg)
        hellinger_loss(target, hist, alpha=2.0).backward()

    def time_kernels(iters, method=None):
        """HIP events (on the stream the kernels are launched on) around the forward and the backward
        C-ABI calls, buffers preallocated: pure device time of [k_hist_fwd + reduce + normalize] and
        of [k_hist_bwd] -- the small kernels are <5 % of either (profiles/).  method: another kernel method on the
        same input (the scatter-add path of 'thresholding')."""
        import ctypes
        from histogan_amd import hist as HH
        from histogan_amd._lib import lib, check
        xd = x.detach()
        p, keep = HH._make_params(xd, cfg if method is None else HistConfig(h=h, insz=insz, method=method, sigma=0.02))
        fb, bb = HH._ws_bytes(p)
        out = torch.empty(B, 3, h, h, device=dev)
        sums = torch.empty(B, device=dev)
        gx = torch.empty_like(xd)
        gout = torch.rand(B, 3, h, h, device=dev) - 0.5
        ws = torch.empty(max(fb, bb, 4), dtype=torch.uint8, device=dev)
        if method is None:      # the forward -> backward projection cache the autograd Function passes (hist.py)
            cache = torch.empty(B, int(p.Hs) * int(p.Ws), 8, device=dev)
            p.proj_cache = cache.data_ptr()
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
        for it in range(iters + 2):
            e = evs[max(it - 2, 0)]
            e[0].record()
            check(lib.hg_rgbuv_hist_fwd(ctypes.byref(p), xd.data_ptr(), out.data_ptr(), sums.data_ptr(),
                                        ws.data_ptr(), ws.numel(), st), 'fwd')
            e[1].record()
            check(lib.hg_rgbuv_hist_bwd(ctypes.byref(p), xd.data_ptr(), gout.data_ptr(), out.data_ptr(),
                                        sums.data_ptr(), gx.data_ptr(), ws.data_ptr(), ws.numel(), st), 'bwd')
            e[2].record()
        torch.cuda.synchronize()
        tf = sum(e[0].elapsed_time(e[1]) for e in evs) / iters * 1e-3
        tb = sum(e[1].elapsed_time(e[2]) for e in evs) / iters * 1e-3
        return tf, tb

    N = min(S, insz) ** 2                       # pixels entering the histogram (the resize is active when S > insz)
    flops_fwd = 6.0 * N * h * h * B            # 3 planes x 2 h^2 flop per pixel
    flops_bwd = 2.0 * flops_fwd
    bytes_fwd = B * (3 * S * S + 3 * h * h) * 4
    bytes_bwd = B * (2 * 3 * S * S + 3 * h * h) * 4
    info = dict(workload=f'rgbuv_hist fwd+hellinger+bwd {B}x3x{S}x{S} h={h} inverse-quadratic sigma=0.02 insz={insz}',
                batch_per_gpu=B, image_size=S, h=h, method='inverse-quadratic', parallelism=f'dp{world}')
    return step, time_kernels, dict(flops_fwd=flops_fwd, flops_bwd=flops_bwd, bytes_fwd=bytes_fwd,
                                    bytes_bwd=bytes_bwd), info, B


def g_layers(size, cap):
    """The generator's 3x3 convolutions (K, N, S) (reference filter arithmetic histoGAN/histoGAN.py:539-543).
    256^2 / capacity 16: (64, 2048, 4), (2048, 2048, 4), (2048, 1024, 8), ... (64, 32, 256), (32, 32, 256)."""
    import math
    L = int(math.log2(size)) - 1
    f = [4 * cap] + [cap * 2 ** (i + 1) for i in range(L)][::-1]
    out = []
    for i in range(L):
        S = 4 * 2 ** i
        out += [(f[i], f[i + 1], S), (f[i + 1], f[i + 1], S)]
    return out


G_LAYERS = g_layers(256, 16)


TRAFFIC_FILE = 'r06_pmc_traffic.json'


def source_digest(name):
    """sha256[:16] of a kernel source file: PMC traffic numbers are only valid for the source they were measured on."""
    import hashlib
    try:
        with open(os.path.join(ROOT, 'histogan_amd', 'csrc', name), 'rb') as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def recorded_traffic(key):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE) from the committed rocprofv3 --pmc passes of exactly this launch
    (profiles/r06_pmc_traffic.json 'bench': tools/wino_pmc.sh / conv_traffic.sh / hist_traffic.sh / make_traffic_record_rounds.py; FETCH_SIZE x 2
    as calibrated by tools/ubench/fetch_calib.hip, profiles/r05_fetch_calibration.txt).  The record carries the digest of
    the kernel source it was measured on: when the source has changed since, the number is stale and None is reported
    (VERDICT r2 weak #8).  Returns (bytes or None, provenance string)."""
    try:
        with open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE)) as f:
            rec = json.load(f)['bench'][key]
    except Exception:
        return None, 'no PMC record for this launch'
    src = rec.get('source')
    if src and rec.get('source_sha16') != source_digest(src):
        return None, f'stale: {src} changed since the PMC pass (recorded at {rec.get("commit", "?")})'
    return rec['fetch_bytes'] + rec['write_bytes'], f'profiles/{TRAFFIC_FILE}, measured at {rec.get("commit", "?")}'


def thr_probe_batch(dev, B, S, h, iters=10):
    """Forward + backward of the thresholding histogram through the C ABI at another batch size (HIP events)."""
    import ctypes
    from histogan_amd import hist as HH
    from histogan_amd._lib import lib, check
    x = torch.rand(B, 3, S, S, device=dev)
    p, keep = HH._make_params(x, HH.HistConfig(h=h, insz=S, method='thresholding'))
    fb, bb = HH._ws_bytes(p)
    out = torch.empty(B, 3, h, h, device=dev); sums = torch.empty(B, device=dev); gx = torch.empty_like(x)
    gout = torch.rand(B, 3, h, h, device=dev) - 0.5
    ws = torch.empty(max(fb, bb, 4), dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    for it in range(iters + 2):
        e = evs[max(it - 2, 0)]
        e[0].record()
        check(lib.hg_rgbuv_hist_fwd(ctypes.byref(p), x.data_ptr(), out.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel(), st), 'fwd')
        e[1].record()
        check(lib.hg_rgbuv_hist_bwd(ctypes.byref(p), x.data_ptr(), gout.data_ptr(), out.data_ptr(), sums.data_ptr(), gx.data_ptr(), ws.data_ptr(), ws.numel(), st), 'bwd')
        e[2].record()
    torch.cuda.synchronize()
    tf = sum(e[0].elapsed_time(e[1]) for e in evs) / iters * 1e-3
    tb = sum(e[1].elapsed_time(e[2]) for e in evs) / iters * 1e-3
    nbytes = B * (3 * S * S + 3 * h * h) * 4 + B * (6 * S * S + 3 * h * h) * 4
    del x, gx, ws
    torch.cuda.empty_cache()
    return {'fwd_ms': tf * 1e3, 'bwd_ms': tb * 1e3, 'achieved': nbytes / (tf + tb) / 1e9, 'unit': 'GB/s',
            'frac': nbytes / (tf + tb) / 1e9 / HBM_PEAK_GBPS, 'bytes': nbytes}


def recorded_value(key):
    """A number taken from a committed rocprofv3 run (profiles/r04_recorded.json, else round 3's), None if not recorded."""
    for name in ('r04_recorded.json', 'r03_recorded.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                return json.load(f)[key]
        except Exception:
            continue
    return None


def _time_calls(calls, iters):
    ts = []
    for fn in calls:
        if fn is None:
            ts.append(None)
            continue
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e-3)
    return ts


def conv_kernel_times(dev, B, iters=6, layers=None):
    """HIP events (torch's current stream == the stream the C ABI launches on) around the C-ABI launches of every generator
    3x3 layer with preallocated buffers: the direct implicit GEMM (hg_conv2d_fwd / _dgrad / _wgrad) and, where the library
    dispatches it (hg_wino_supported / hg_wino_wgrad_supported), the Winograd form (hg_wino_conv2d / hg_wino_wgrad).
    Returns {layer: dict(flops, direct=(t_fwd, t_dgrad, t_wgrad), wino=(t or None, ...))} in seconds per launch."""
    import ctypes
    from histogan_amd import conv as C
    from histogan_amd._lib import lib, check
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for K, N, S in (layers or G_LAYERS):
        x = torch.randn(B, K, S, S, device=dev)
        w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
        go = torch.randn(B, N, S, S, device=dev)
        y, gx, gw = torch.empty_like(go), torch.empty_like(x), torch.empty_like(w)
        wf, wd = C._pack_weights(w, C.PACK_FWD), C._pack_weights(w, C.PACK_DGRAD)
        nf = lib.hg_conv2d_workspace_bytes(B, K, N, S, S, 3, 1, 0)
        nd = lib.hg_conv2d_workspace_bytes(B, N, K, S, S, 3, 1, 1)
        nw = lib.hg_conv2d_wgrad_workspace_bytes(B, K, N, S, S, 3, 1)
        wnf, wnd = lib.hg_wino_workspace_bytes(B, K, N, S, S), lib.hg_wino_workspace_bytes(B, N, K, S, S)
        wnw = lib.hg_wino_wgrad_workspace_bytes(B, K, N, S, S)
        ws = torch.empty(max(nf, nd, nw, wnf, wnd, wnw, 4), dtype=torch.uint8, device=dev)
        direct = _time_calls((
            lambda: check(lib.hg_conv2d_fwd(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, None, None, B, K, N, S, S,
                                            3, 1, ws.data_ptr(), ws.numel(), st), 'fwd'),
            lambda: check(lib.hg_conv2d_dgrad(go.data_ptr(), wd.data_ptr(), gx.data_ptr(), None, None, B, N, K, S, S,
                                              3, 1, ws.data_ptr(), ws.numel(), st), 'dgrad'),
            lambda: check(lib.hg_conv2d_wgrad(x.data_ptr(), go.data_ptr(), gw.data_ptr(), None, None, B, K, N, S, S,
                                              3, 1, ws.data_ptr(), ws.numel(), st), 'wgrad')), iters)
        uf = C._wino_pack(w, C.PACK_FWD) if C.wino_supported(B, K, N, S, S) else False
        ud = C._wino_pack(w, C.PACK_DGRAD) if C.wino_supported(B, N, K, S, S) else False
        wino = _time_calls((
            (lambda: check(lib.hg_wino_conv2d(x.data_ptr(), uf.data_ptr(), y.data_ptr(), None, None, None, None, None, 0, 0.0,
                                              None, B, K, N, S, S, ws.data_ptr(), ws.numel(), st), 'wino fwd')) if uf is not False else None,
            (lambda: check(lib.hg_wino_conv2d(go.data_ptr(), ud.data_ptr(), gx.data_ptr(), None, None, None, None, None, 0, 0.0,
                                              None, B, N, K, S, S, ws.data_ptr(), ws.numel(), st), 'wino dgrad')) if ud is not False else None,
            (lambda: check(lib.hg_wino_wgrad(x.data_ptr(), go.data_ptr(), gw.data_ptr(), B, K, N, S, S, ws.data_ptr(), ws.numel(),
                                             st), 'wino wgrad')) if C.wino_wgrad_supported(B, K, N, S, S) else None), iters)
        out[(K, N, S)] = dict(flops=2.0 * B * S * S * K * N * 9, direct=tuple(direct), wino=tuple(wino))
    return out


WINO_FLOP_RATIO = 16.0 / 36.0   # executed multiplications of F(2x2,3x3) per multiplication of the direct 3x3 convolution


def wino_line(kernel, direct_flops, t, extra=None):
    """A roofline entry of a Winograd launch: `achieved` / `frac` count the MFMA flops the kernel EXECUTES (16 products per
    2x2 output tile, input and output channel); what a direct convolution would have to sustain for the same launch time is
    the separate key `direct_equivalent_tflops` -- a speed-up measure, not a roofline fraction."""
    ex = direct_flops * WINO_FLOP_RATIO
    d = {'kernel': kernel, 'bound': 'mfma', 'achieved': ex / t / 1e12, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
         'frac': ex / t / 1e12 / FP32_PEAK_TFLOPS, 'launch_ms': t * 1e3, 'flops_per_launch': ex,
         'direct_equivalent_tflops': direct_flops / t / 1e12, 'direct_conv_flops_per_launch': direct_flops}
    if extra:
        d.update(extra)
    return d


def leading_kernel_lines(dev, B, ct, size, cap):
    """Stand-alone lines for the other launches that lead the step's kernel trace next to the headline roofline launch:
    (a) the Winograd data gradient at 512 -> 256 channels, 32 x 32 (hg_wino_conv2d on the transposed / flipped operand);
    (b) the no-autograd generator stage (ops.modconv_stage: modulation while the patches are loaded, convolution,
    demodulation, noise, LeakyReLU in ONE hg_wino_conv2d launch, plus its demodulation-coefficient launches) at the roofline
    layer; (c) the DIRECT k_conv at 512 -> 256, 32 x 32, the kernel that still runs every 1x1 / stride-2 / few-channel layer."""
    from histogan_amd import ops
    out = []
    key = (32 * cap, 16 * cap, size // 8)
    if key in ct:
        fl = ct[key]['flops']
        if ct[key]['wino'][1] is not None:
            out.append(wino_line('k_wino (hg_wino_conv2d, data gradient) at %d->%d ch, %dx%d, batch %d' % (key[1], key[0], key[2], key[2], B),
                                 fl, ct[key]['wino'][1]))
        tf = ct[key]['direct'][0]
        out.append({'kernel': 'k_conv<KC=4, FE=false> (hg_conv2d_fwd, direct implicit GEMM) at %d->%d ch, %dx%d, batch %d' % (*key, key[2], B), 'bound': 'mfma',
                    'achieved': fl / tf / 1e12, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': fl / tf / 1e12 / FP32_PEAK_TFLOPS,
                    'launch_ms': tf * 1e3})
    K, N, S = 16 * cap, 8 * cap, size // 4
    with torch.no_grad():
        x = torch.randn(B, K, S, S, device=dev)
        st = 0.3 * torch.randn(B, K, device=dev)
        w = torch.randn(N, K, 3, 3, device=dev) / (K * 9) ** 0.5
        nzt = torch.rand(B, size, size, device=dev)
        wn, bn = torch.randn(N, device=dev), torch.randn(N, device=dev)
        fn = lambda: ops.modconv_stage(x, st, w, nzt, wn, bn, demod=True, upsample=False, act=True)
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 6 * 1e-3
    fl = 2.0 * B * S * S * K * N * 9
    from histogan_amd import conv as C
    name = ('generator stage (ops.modconv_stage: modulate + conv + demodulate + noise + LeakyReLU, incl. its demodulation-'
            'coefficient launches) at %d->%d ch, %dx%d, batch %d' % (K, N, S, S, B))
    if C.wino_supported(B, K, N, S, S):
        out.append(wino_line('k_wino<FE=true> (hg_wino_conv2d) ' + name, fl, t))
    else:
        out.append({'kernel': 'k_conv<FE=true> (hg_modconv2d_fwd) ' + name, 'bound': 'mfma', 'achieved': fl / t / 1e12,
                    'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': fl / t / 1e12 / FP32_PEAK_TFLOPS, 'launch_ms': t * 1e3})
    return out


def oracle_train_step(args, n, device):
    """The reference's op chain for one train step (oracle/: functional restatement of its networks + histogram block,
    stock aten ops) on `device`, for n images -- D phase: G fwd (no grad), D on fake and real, backward; G phase: G fwd,
    D fwd, RGB-uv histogram + Hellinger loss, backward.  Returns (one_step, diffgrad_once, n_params)."""
    import torch.nn.functional as F
    from histogan_amd.nets import Discriminator, Generator, HistVectorizer, StyleVectorizer
    from oracle import histogan_nets as N
    from oracle import rgbuv_hist as O
    S, cap, h = args.size, args.capacity, args.bins
    L = int(__import__('math').log2(S)) - 1
    torch.manual_seed(0)
    sd = {k: {kk: v.detach().to(device).requires_grad_(True) for kk, v in m.state_dict(keep_vars=True).items()}
          for k, m in dict(G=Generator(S, 512, cap), D=Discriminator(S, cap), S=StyleVectorizer(512, 8),
                           H=HistVectorizer(h, 512, 8)).items()}
    img = torch.rand(n, 3, S, S, device=device)
    tgt = O.rgbuv_hist(torch.rand(n, 3, S, S, device=device), h=h, insz=150)

    def gen(grad):
        with torch.set_grad_enabled(grad):
            w = N.vectorizer(sd['S'], torch.randn(n, 512, device=device), 'net')
            hw = N.vectorizer(sd['H'], tgt, 'fcs')
            return N.generator(sd['G'], w[:, None].expand(-1, L - 2, -1), hw[:, None].expand(-1, 2, -1),
                               torch.rand(n, S, S, 1, device=device), L)

    def one_step():
        fake = gen(False)
        real = img.clone().requires_grad_(True)
        d = (F.relu(1 + N.discriminator(sd['D'], real, L + 1)) + F.relu(1 - N.discriminator(sd['D'], fake, L + 1))).mean()
        torch.autograd.grad(d, list(sd['D'].values()), allow_unused=True)
        fake = gen(True)
        loss = N.discriminator(sd['D'], fake, L + 1).mean() + O.hellinger_loss(tgt, O.rgbuv_hist(F.relu(fake), h=h, insz=150), 2.0)
        torch.autograd.grad(loss, [v for k in 'GSH' for v in sd[k].values()], allow_unused=True)

    flat = torch.cat([v.detach().reshape(-1) for k in 'GDSH' for v in sd[k].values()])
    state = dict(step=0, exp_avg=torch.zeros_like(flat), exp_avg_sq=torch.zeros_like(flat),
                 previous_grad=torch.zeros_like(flat))

    def diffgrad_once():
        N.diffgrad_step(flat, torch.randn_like(flat), state, 2e-4)

    return one_step, diffgrad_once, flat.numel()


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _median_time(fn, reps=3, sync=None):
    """One warm-up call, then the median wall time of `reps` calls."""
    fn()
    ts = []
    for _ in range(reps):
        if sync:
            sync()
        t0 = time.perf_counter()
        fn()
        if sync:
            sync()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def cpu_baseline_train(args):
    """The oracle doing the compute of one train step on the host cores, on a bounded sample of the batch: one warm-up,
    then the median of three timings (torch CPU, all threads)."""
    n = args.cpu_images
    one_step, diffgrad_once, nparam = oracle_train_step(args, n, torch.device('cpu'))
    dt = _median_time(one_step, reps=args.cpu_reps)
    dto = _median_time(diffgrad_once, reps=args.cpu_reps)   # optimizer: DiffGrad over every parameter once per step (amortised over the batch)
    per_img = dt / n + dto / args.batch
    return dict(value=1.0 / per_img, unit='images/s', cores=torch.get_num_threads(), kind='port', cpu=cpu_model(),
                sample=f'{n} of the {args.batch} images: one D phase + one G phase (no GP / PL step) {dt:.1f} s, '
                       f'DiffGrad over {nparam/1e6:.0f} M parameters {dto:.2f} s amortised over the batch; '
                       f'torch CPU {torch.get_num_threads()} threads; 1 warm-up + median of {args.cpu_reps}')


def reference_eager_rocm(args, dev):
    """'Just run the repo on AMD' (BASELINE.md section 3.4): the reference's op chain -- per-sample modulated weights +
    one grouped F.conv2d, nn.Upsample, the ~80-launch-per-image histogram chain with fp64 temporaries -- executed by
    stock PyTorch-ROCm eager (MIOpen / rocBLAS / aten) on the same GPU, same batch, fp32.  Outside the timed region;
    the operator chain is the oracle's restatement (the reference itself is not on the GPU box)."""
    n = args.batch
    try:
        one_step, diffgrad_once, nparam = oracle_train_step(args, n, dev)
        dt = _median_time(one_step, reps=3, sync=torch.cuda.synchronize)
        dto = _median_time(diffgrad_once, reps=3, sync=torch.cuda.synchronize)
        torch.cuda.empty_cache()
        return dict(value=n / (dt + dto), unit='images/s', ms_per_step=(dt + dto) * 1e3, batch=n, kind='port',
                    sample=f'one plain D+G step (no GP / PL) at batch {n}: {dt*1e3:.0f} ms + DiffGrad (aten, {nparam/1e6:.0f} M '
                           f'parameters) {dto*1e3:.0f} ms; 1 warm-up + median of 3; aten/MIOpen/rocBLAS eager, fp32')
    except Exception as e:      # e.g. out of memory for the materialised per-sample weights at a larger size
        torch.cuda.empty_cache()
        return dict(value=None, error=f'{type(e).__name__}: {str(e)[:200]}')


def train_workload(args, dev, rank, world):
    from histoGAN import Trainer
    tr = Trainer('bench', '/tmp/hg_bench_results', '/tmp/hg_bench_models', args.size, args.capacity,
                 batch_size=args.batch, hist_bin=args.bins, hist_insz=150, hist_resizing='interpolation',
                 hist_method='inverse-quadratic', hist_sigma=0.02, attn_layers=list(args.attn_layers))
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    tr.init_GAN()

    def step():
        tr.train(alpha=2)
    step.trainer = tr

    info = dict(workload=f'HistoGAN G+D train step {args.size}x{args.size} capacity={args.capacity} '
                         f'batch={args.batch}/GPU h={args.bins} insz=150 inverse-quadratic (GP every 4th, PL every 32nd step)'
                         + (f' attn_layers={list(args.attn_layers)}' if args.attn_layers else ''),
                batch_per_gpu=args.batch, global_batch=args.batch * world, image_size=args.size,
                network_capacity=args.capacity, h=args.bins, parallelism=f'dp{world}',
                params_G=sum(p.numel() for p in tr.GAN.G.parameters()),
                params_D=sum(p.numel() for p in tr.GAN.D.parameters()))
    return step, info, args.batch


def rehistogan_workload(args, dev, rank, world):
    """ReHistoGAN train step: recoloringTrainer.train on resident synthetic batches."""
    from ReHistoGAN import recoloringTrainer
    tr = recoloringTrainer('bench_re', '/tmp/hg_bench_results', '/tmp/hg_bench_models', args.size, args.capacity,
                           batch_size=args.batch, hist_bin=args.bins, hist_insz=150, hist_resizing='interpolation',
                           rec_loss='laplacian', variance_loss=False)
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    tr.init_GAN()

    def step():
        tr.train(alpha=32, beta=1.5, gamma=4)
    step.trainer = tr

    n = lambda m: sum(p.numel() for p in m.parameters())
    info = dict(workload=f'ReHistoGAN recolouring train step {args.size}x{args.size} capacity={args.capacity} '
                         f'batch={args.batch}/GPU h={args.bins} Laplacian reconstruction loss (GP every 4th step)',
                batch_per_gpu=args.batch, global_batch=args.batch * world, image_size=args.size,
                network_capacity=args.capacity, h=args.bins, parallelism=f'dp{world}',
                params_ED=n(tr.GAN.ED), params_G=n(tr.GAN.G), params_D=n(tr.GAN.D))
    return step, info, args.batch


def cpu_baseline(args):
    """The oracle (port of the reference's PyTorch CPU path) on a bounded sample of the workload: one warm-up, median of 3."""
    from oracle import rgbuv_hist as O
    S, h = args.size, args.bins
    nimg = args.cpu_images
    g = torch.Generator().manual_seed(5)
    x = torch.rand(nimg, 3, S, S, generator=g)
    tgt = O.rgbuv_hist(torch.rand(nimg, 3, S, S, generator=g), h=h, insz=S)
    dt = _median_time(lambda: O.rgbuv_hist_fwd_bwd(x, target=tgt, alpha=2.0, h=h, insz=S), reps=args.cpu_reps)
    return dict(value=nimg / dt, unit='images/s', cores=torch.get_num_threads(), kind='port', cpu=cpu_model(),
                sample=f'{nimg} of the {args.batch} images ({nimg}x3x{S}x{S}), fwd+Hellinger+bwd, torch CPU '
                       f'{torch.get_num_threads()} threads, {dt:.2f} s; 1 warm-up + median of {args.cpu_reps}')


def cpu_baseline_hist_c1(args):
    """configs[0] of BASELINE.json (C1): RGBuvHistBlock forward + Hellinger loss on 4 x 3 x 128 x 128 random RGB, the reference
    notebook's CPU path (Histogram_loss.ipynb:394-417; h = 64, insz = 150: no resize at 128), on the oracle (the restated
    PyTorch CPU chain), all host threads.  Forward + loss is the notebook's case; forward + loss + backward beside it."""
    from oracle import rgbuv_hist as O
    g = torch.Generator().manual_seed(0)
    x = torch.rand(4, 3, 128, 128, generator=g)
    tgt = O.rgbuv_hist(torch.rand(4, 3, 128, 128, generator=torch.Generator().manual_seed(1)), h=64, insz=150)

    def fwd_loss():
        with torch.no_grad():
            return O.hellinger_loss(tgt, O.rgbuv_hist(x, h=64, insz=150))
    dt_f = _median_time(fwd_loss, reps=5)
    dt_fb = _median_time(lambda: O.rgbuv_hist_fwd_bwd(x, target=tgt, alpha=1.0, h=64, insz=150), reps=5)
    return dict(value=4 / dt_f, unit='images/s', cores=torch.get_num_threads(), kind='port', cpu=cpu_model(),
                ms_fwd_loss=dt_f * 1e3, ms_fwd_loss_bwd=dt_fb * 1e3, images_per_s_fwd_loss_bwd=4 / dt_fb,
                sample='the whole C1 case: 4x3x128x128, h=64, inverse-quadratic; 1 warm-up + median of 5')


def ddp_probe(dist, dev, rank, world, tr):
    """Self-check of the data-parallel set-up for the driver's SCALE record: every rank reports the world size / backend
    it sees and its device, and the gradient all-reduces of one step are timed stand-alone (outside the timed region):
    payload bytes, ms, algorithm bandwidth and bus bandwidth (2 (N-1)/N x payload / time) per collective."""
    backend = dist.get_backend()
    seen = [None] * world
    dist.all_gather_object(seen, dict(rank=rank, world_size=dist.get_world_size(), device=str(dev),
                                      device_name=torch.cuda.get_device_name(dev)))
    info = {'backend': backend, 'ranks': seen, 'hellinger': 'global batch (one scalar all-reduce per step)',
            'grad_allreduce_op': 'AVG in the collective' if backend == 'nccl' else 'pre-scale + SUM'}
    if backend == 'nccl':           # one process per GPU: every rank must sit on its own device
        assert len({r['device'] for r in seen}) == world and all(r['world_size'] == world for r in seen), seen
    if tr is None or getattr(tr, 'GAN', None) is None:
        return info
    res = {}
    for name, flat in (('D', tr.GAN._flat_d), ('G+S+H', tr.GAN._flat_g)):
        buf = torch.zeros_like(flat.grad)
        for _ in range(2):
            dist.all_reduce(buf)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        nbytes = buf.numel() * 4
        res[name] = {'bytes': nbytes, 'ms': ms, 'algbw_GBps': nbytes / ms / 1e6,
                     'busbw_GBps': 2.0 * (world - 1) / world * nbytes / ms / 1e6}
        del buf
    info['allreduce'] = res
    info['allreduce_ms_per_step'] = sum(v['ms'] for v in res.values())
    return info


def claim_stdout():
    """The contract is ONE JSON line on stdout.  librccl writes a version banner to the C stdout when the process group is
    torn down (5 lines behind the JSON at any world size), and a stray library print would do the same: file descriptor 1
    is pointed at stderr for everything in this process, and the JSON line goes to the original stdout."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(saved, 'w')


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_env(base, rank, world, port):
    """Environment of rank `rank` of a `world`-rank single-node job (what torch.distributed.run would export)."""
    env = dict(base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('HG_BENCH_LAUNCHED', None)
    env['HG_BENCH_LAUNCHED'] = '1'
    return env


def launch_ranks(n, argv, json_out, timeout=None):
    """`python bench.py --gpus N` without an external launcher: start one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as torch.distributed.run exports them), forward rank 0's single JSON line, return the worst exit code.
    One rank per device is asserted up front (HG_DIST_BACKEND=gloo: ranks may share a device -- test use)."""
    import subprocess
    backend = os.environ.get('HG_DIST_BACKEND', 'nccl')
    have = torch.cuda.device_count()
    if backend == 'nccl' and have < n:
        print(f'bench.py: --gpus {n} but only {have} GPU(s) visible (one process per GPU over RCCL; '
              f'HG_DIST_BACKEND=gloo lets ranks share a device for tests)', file=sys.stderr)
        return 2
    port = free_port()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=rank_env(os.environ, r, n, port),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    # Supervise EVERY rank: when one exits non-zero (out of memory, bad device) the others would sit in the rendezvous or a
    # collective for ever -- they are killed and its exit code returned (what torch.distributed.run does for its workers).
    import threading
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    t_end = time.time() + (timeout if timeout is not None else 3600.0)
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > t_end:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:              # exactly the processes started here
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    rd.join(timeout=10)
    out0 = b''.join(c for c in chunks if c)
    lines = [l for l in out0.decode(errors='replace').splitlines() if l.strip()]
    if rc == 0 and lines:
        print(lines[-1], file=json_out, flush=True)
    return rc


def main():
    json_out = claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=32)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--workload', default='train', choices=['train', 'hist', 'rehistogan', 'c5'],
                    help="c5 = BASELINE.json configs[4] on one GPU: train step at 1024^2, capacity 16, batch 8, h = 128, attention")
    ap.add_argument('--attn-layers', type=lambda v: [int(t) for t in v.split(',') if t], default=[])
    ap.add_argument('--hist-insz', type=int, default=0, help='histogram input size of the stand-alone histogram timings (0: image size)')
    ap.add_argument('--capacity', type=int, default=16)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--bins', type=int, default=64)
    ap.add_argument('--cpu-images', type=int, default=4)
    ap.add_argument('--cpu-reps', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-reference-eager', action='store_true')
    ap.add_argument('--no-roofline', action='store_true', help='skip the stand-alone kernel timings (tests only)')
    ap.add_argument('--no-alt-precision', action='store_true', help='(accepted and ignored: the bf16x6 secondary line was removed in round 6)')
    args = ap.parse_args()

    if args.workload == 'c5':
        # BASELINE.json configs[4]: HistoGAN 1024^2 (capacity 16, discriminator attention on), batch 8 per GPU, h = 128
        # inverse-quadratic.  attn_layers = [3, 4] (128^2 and 64^2 maps): attention after block 1 would keep q / k / v of 512
        # channels on 512^2 maps -- 8.6 GB each for the 16-image [fake; real] pass -- and does not fit 288 GB together with the
        # gradient penalty's double backward.  A secondary workload: the headline stays configs[2].
        args.workload, args.size, args.capacity, args.batch, args.bins = 'train', 1024, 16, 8, 128
        args.attn_layers = args.attn_layers or [3, 4]
        args.hist_insz = 150
        args.no_cpu_baseline = args.no_reference_eager = True      # (the oracle needs minutes per 1024^2 image on the host)
        args.c5 = True
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no external launcher (torch.distributed.run exports WORLD_SIZE): one process per GPU from here
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], json_out))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dev = torch.device('cuda', local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or os.environ.get('HG_DIST_FORCE', '0') == '1':    # (forced at world size 1: RCCL exercised on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # 'nccl' == RCCL over xGMI.  HG_DIST_BACKEND=gloo exists to exercise the N>1 path on a box with fewer GPUs than
        # ranks (ranks then share a device; test use only)
        backend = os.environ.get('HG_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            try:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
            except TypeError:        # older torch: no device_id argument
                dist.init_process_group('nccl', rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    hstep, time_kernels, work, hinfo, units = hist_workload(args, dev, rank, world)
    if args.workload == 'train':
        step, info, units = train_workload(args, dev, rank, world)
    elif args.workload == 'rehistogan':
        step, info, units = rehistogan_workload(args, dev, rank, world)
    else:
        step, info = hstep, hinfo

    ddp_info = None
    if dist:
        ddp_info = ddp_probe(dist, dev, rank, world, getattr(step, 'trainer', None))
    for _ in range(args.warmup):
        step()
    tr = getattr(step, 'trainer', None)
    if tr is not None and hasattr(tr, '_graph_eligible'):
        # HG_GRAPH=auto decides from its first eager plain steps whether plain / gradient-penalty steps replay from
        # captured hipGraphs; a capture is a one-time ~0.15 s.  Keep both out of the timed window: untimed settle steps
        # until the decision is made and, if it is "graph", both graphs exist (at most 16 steps).
        for _ in range(16):
            decided = tr.graph_mode != 'auto' or getattr(tr, '_graph_auto', None) is not None
            use = tr.graph_mode == '1' or getattr(tr, '_graph_auto', False)
            have = {k[0] for k in getattr(tr, '_graphs', {}).keys()}
            if decided and (not use or getattr(tr, '_graph_failed', False) or have >= {False, True}):
                break
            step()
    if tr is not None and hasattr(tr, 'pl_mean'):
        # One untimed step of the path-length kind with the running mean set (step 0 of the warm-up has none and skips
        # that loss term): the first such step grows the allocator for the backward through two generator passes
        # (measured: 159 ms instead of 85 ms when it fell into the timed window).
        tr.steps = 32 * ((tr.steps + 31) // 32)
        step()
    if tr is not None:
        # Pin the schedule phase: the timed window starts on a step with steps % 32 == 0, so K timed steps always hold
        # ceil(K/4) gradient-penalty steps and ceil(K/32) path-length steps (the reference's mix, histoGAN.py:882-883),
        # whatever --warmup was.  (With K < 32 the one path-length step weighs more than its 1/32 share: the
        # `schedule_mix` entry below re-weights the measured per-kind means to the 32-step period.)
        # Windows that are not a multiple of 32 hold round(K / 32) path-length steps (K = 20: one; K = 8: none, the
        # window then starts one step later) -- the composition closest to the 32-step period.
        tr.steps = 32 * ((tr.steps + 31) // 32) + (0 if (args.steps % 32 == 0 or round(args.steps / 32.0) >= 1) else 1)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    per_step, kinds, host_ms, graphed = [], [], [], []
    ev0 = None
    if tr is not None and hasattr(tr, 'keep_step_events'):
        tr.keep_step_events, tr.step_events = True, []
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        if tr is not None:
            kinds.append('gp+pl' if tr.steps % 32 == 0 else ('gp' if tr.steps % 4 == 0 else 'plain'))
        step()                      # (train() ends with its one blocking read-back: the wall time of a call is the step time)
        per_step.append(time.perf_counter() - ts)
        if tr is not None:
            host_ms.append(getattr(tr, 'host_enqueue_ms', float('nan')))
            graphed.append(bool(getattr(tr, 'last_step_graphed', False)))
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ev0 is not None and len(tr.step_events) == args.steps:
        # per-step GPU time between the end-of-step events on the main stream (with the deferred read-back the host's
        # wall time of call i covers the completion of step i-1, not of step i)
        evs = [ev0] + [e for _, e in tr.step_events]
        per_step = [evs[i].elapsed_time(evs[i + 1]) * 1e-3 for i in range(args.steps)]
        tr.keep_step_events = False
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if args.no_roofline:
        if rank == 0:
            print(json.dumps({'metric': METRIC, 'value': units * world * args.steps / dt, 'unit': 'images/s', 'n_gpus': world,
                              'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
                              'config': info, 'ddp': ddp_info, 'roofline': None,
                              'graph_replayed_steps': int(sum(graphed))}), file=json_out, flush=True)
        if dist:
            dist.destroy_process_group()
        return
    t_fwd, t_bwd = time_kernels(min(max(args.steps, 5), 20))
    hist_roof = {'kernel': 'k_hist_bwd', 'bound': 'mfma', 'achieved': work['flops_bwd'] / t_bwd / 1e12,
                 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': work['flops_bwd'] / t_bwd / 1e12 / FP32_PEAK_TFLOPS,
                 'traffic': recorded_traffic('k_hist_bwd_c2')[0] if (args.batch, args.size, args.bins) == (32, 256, 64) else None,
                 'traffic_source': recorded_traffic('k_hist_bwd_c2')[1],
                 'launch_ms': t_bwd * 1e3,
                 'fwd': {'kernel': 'k_hist_fwd', 'achieved': work['flops_fwd'] / t_fwd / 1e12,
                         'frac': work['flops_fwd'] / t_fwd / 1e12 / FP32_PEAK_TFLOPS, 'launch_ms': t_fwd * 1e3}}
    # the HBM-side method of the same block: thresholding on the scatter-add / gather kernels
    tt_f, tt_b = time_kernels(min(max(args.steps, 5), 20), 'thresholding')
    thr_gbps = (work['bytes_fwd'] + work['bytes_bwd']) / (tt_f + tt_b) / 1e9
    hist_roof['thresholding'] = {'kernels': 'k_thr_fwd_lean + k_hist_finish / k_thr_bwd_lean', 'bound': 'hbm',
                                 'fwd_ms': tt_f * 1e3, 'bwd_ms': tt_b * 1e3, 'achieved': thr_gbps, 'peak': HBM_PEAK_GBPS,
                                 'unit': 'GB/s', 'frac': thr_gbps / HBM_PEAK_GBPS,
                                 'traffic': recorded_traffic('thr_fwd_bwd_c2')[0] if (args.batch, args.size, args.bins) == (32, 256, 64) else None,
                                 'traffic_source': recorded_traffic('thr_fwd_bwd_c2')[1],
                                 'north_star_hbm_target': 0.6, 'north_star_hbm_target_met': bool(thr_gbps / HBM_PEAK_GBPS >= 0.6),
                                 'note': 'three launches of 17 + 5 + 23 us at batch 32: launch-latency regime (78.6 MB = 9.8 us at peak)'}
    if rank == 0 and world == 1 and (args.size, args.bins) == (256, 64) and args.batch < 256:
        # the same kernels where the launches are long enough to stream: batch 256 (629 MB per forward + backward)
        hist_roof['thresholding']['batch256'] = thr_probe_batch(dev, 256, args.size, args.bins)
    if args.workload in ('train', 'rehistogan'):
        ct = conv_kernel_times(dev, args.batch, layers=g_layers(args.size, args.capacity))
        rl = (16 * args.capacity, 8 * args.capacity, args.size // 4)        # 256 -> 128 channels at a quarter of the image size
        fl = ct[rl]['flops']
        tf, td, tw = ct[rl]['direct']
        wf_, wd_, ww_ = ct[rl]['wino']
        where = 'at %d->%d ch, %dx%d, batch %d' % (rl[0], rl[1], rl[2], rl[2], args.batch)
        direct_line = {'kernel': 'k_conv<128ch x 128px tile, 3x3, stride 1> (hg_conv2d_fwd, direct implicit GEMM: the kernel the '
                                 'Winograd form replaced on this layer) ' + where, 'bound': 'mfma',
                       'achieved': fl / tf / 1e12, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                       'frac': fl / tf / 1e12 / FP32_PEAK_TFLOPS, 'launch_ms': tf * 1e3, 'flops_per_launch': fl,
                       'traffic': recorded_traffic('k_conv_fwd_256_128_64_b32')[0] if (args.batch, rl) == (32, (256, 128, 64)) else None,
                       'traffic_source': recorded_traffic('k_conv_fwd_256_128_64_b32')[1],
                       'wgrad': {'kernel': 'k_wgrad (hg_conv2d_wgrad), same layer', 'achieved': fl / tw / 1e12,
                                 'frac': fl / tw / 1e12 / FP32_PEAK_TFLOPS, 'launch_ms': tw * 1e3,
                                 'traffic': recorded_traffic('k_wgrad_256_128_64_b32')[0] if (args.batch, rl) == (32, (256, 128, 64)) else None}}
        alg_bytes = 4.0 * (args.batch * rl[0] * rl[2] ** 2 + args.batch * rl[1] * rl[2] ** 2 + 16 * rl[0] * rl[1])
        if wf_ is not None:
            roof = wino_line('k_wino<64 ch x 64 tiles x 16 positions> (hg_wino_conv2d: Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32) ' + where,
                             fl, wf_, {
                'traffic': recorded_traffic('k_wino_fwd_256_128_64_b32')[0] if (args.batch, rl) == (32, (256, 128, 64)) else None,
                'traffic_unit': 'bytes/launch (FETCH_SIZE + WRITE_SIZE)',
                'traffic_source': recorded_traffic('k_wino_fwd_256_128_64_b32')[1],
                'algorithmic_bytes_per_launch': alg_bytes})
            if ww_ is not None:
                roof['wgrad'] = wino_line('k_wino_wgrad + k_wino_wgrad_reduce (hg_wino_wgrad), same layer', fl, ww_, {
                    'traffic': recorded_traffic('k_wino_wgrad_256_128_64_b32')[0] if (args.batch, rl) == (32, (256, 128, 64)) else None,
                    'traffic_unit': 'bytes/launch of k_wino_wgrad alone (the reduce reads its 67 MB of slabs once more)'})
            if wd_ is not None:
                roof['dgrad'] = wino_line('k_wino (hg_wino_conv2d on the data-gradient operand), same layer', fl, wd_)
            roof['direct_kernel'] = direct_line
        else:
            roof = direct_line
            roof['algorithmic_bytes_per_launch'] = alg_bytes
        # the generator's fourteen 3x3 layers as the library dispatches them (Winograd where hg_wino_*_supported, else direct)
        disp = [sum((v['wino'][i] if v['wino'][i] is not None else v['direct'][i]) for v in ct.values()) for i in range(3)]
        dire = [sum(v['direct'][i] for v in ct.values()) for i in range(3)]
        flp = sum(v['flops'] for v in ct.values())
        roof['generator_3x3_layers'] = {
            'direct_conv_flops_per_pass': flp,
            'as_dispatched': {'fwd_ms': disp[0] * 1e3, 'dgrad_ms': disp[1] * 1e3, 'wgrad_ms': disp[2] * 1e3,
                              'fwd_direct_equivalent_tflops': flp / disp[0] / 1e12, 'dgrad_direct_equivalent_tflops': flp / disp[1] / 1e12,
                              'wgrad_direct_equivalent_tflops': flp / disp[2] / 1e12,
                              'layers_on_winograd': {'fwd': sum(v['wino'][0] is not None for v in ct.values()),
                                                     'dgrad': sum(v['wino'][1] is not None for v in ct.values()),
                                                     'wgrad': sum(v['wino'][2] is not None for v in ct.values()), 'of': len(ct)}},
            'direct_kernels': {'fwd_ms': dire[0] * 1e3, 'dgrad_ms': dire[1] * 1e3, 'wgrad_ms': dire[2] * 1e3,
                               'fwd_tflops': flp / dire[0] / 1e12, 'dgrad_tflops': flp / dire[1] / 1e12,
                               'wgrad_tflops': flp / dire[2] / 1e12}}
        roof['hist'] = hist_roof
        try:
            roof['leading_kernels'] = leading_kernel_lines(dev, args.batch, ct, args.size, args.capacity)
        except Exception as e:
            roof['leading_kernels'] = {'error': f'{type(e).__name__}: {str(e)[:160]}'}
    else:
        roof = hist_roof

    if rank == 0:
        out = {
            'metric': METRIC, 'value': units * world * args.steps / dt, 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': info,
            'roofline': roof,
            'hist_hbm_gbps_algorithmic': (work['bytes_fwd'] + work['bytes_bwd']) / (t_fwd + t_bwd) / 1e9,
            'hist_hbm_frac_of_peak': (work['bytes_fwd'] + work['bytes_bwd']) / (t_fwd + t_bwd) / 1e9 / HBM_PEAK_GBPS,
        }
        if kinds:
            mean = lambda k: (sum(t for t, kk in zip(per_step, kinds) if kk == k) / max(1, kinds.count(k))) * 1e3
            sched = {'gp_steps': sum(k != 'plain' for k in kinds), 'pl_steps': kinds.count('gp+pl'),
                     'plain_steps': kinds.count('plain'),
                     'ms_plain': mean('plain') if 'plain' in kinds else None, 'ms_gp': mean('gp') if 'gp' in kinds else None,
                     'ms_gp_pl': mean('gp+pl') if 'gp+pl' in kinds else None}
            if all(sched[k] is not None for k in ('ms_plain', 'ms_gp', 'ms_gp_pl')):
                period = (24 * sched['ms_plain'] + 7 * sched['ms_gp'] + sched['ms_gp_pl']) / 32.0
                sched['ms_per_step_32_period'] = period
                sched['images_per_s_32_period'] = units * world / period * 1e3
            out['schedule_mix'] = sched
            plain_host = [h for h, k in zip(host_ms, kinds) if k == 'plain']
            out['host'] = {'enqueue_ms_plain_step': sum(plain_host) / max(1, len(plain_host)),
                           'enqueue_ms_mean': sum(host_ms) / max(1, len(host_ms)),
                           'graph_mode': getattr(tr, 'graph_mode', None),
                           'deferred_readback': bool(getattr(tr, 'lazy_stats', False)),
                           'graph_replayed_steps': int(sum(graphed)),
                           'launches_per_plain_step_eager': recorded_value('launches_per_plain_step'),
                           'host_calls_per_graphed_step': 1 if any(graphed) else None}
        if dist:
            out['ddp'] = ddp_info
        if world == 1 and not args.no_cpu_baseline:
            if args.workload != 'rehistogan':      # the CPU baseline is quoted for the headline workloads only
                out['cpu_baseline'] = cpu_baseline_train(args) if args.workload == 'train' else cpu_baseline(args)
                out['cpu_baseline_hist_c1'] = cpu_baseline_hist_c1(args)
            if args.workload == 'train' and not args.no_reference_eager:
                out['reference_eager_rocm'] = reference_eager_rocm(args, dev)
        print(json.dumps(out), file=json_out, flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
