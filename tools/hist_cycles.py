#!/usr/bin/env python3
"""Where the shader cycles of the dense histogram kernels go.  Needs the probe build:

    HG_LIB_TAG=probe HG_CFLAGS='-DHG_HIST_PROBE=1' python -m histogan_amd.build      # here, no GPU needed
    HG_LIB_TAG=probe python tools/hist_cycles.py                                      # on the GPU box

k_hist_fwd / k_hist_bwd bracket their phases with s_memtime (shader cycles) and add them up over all waves; this script
runs configs[1] (32 x 3 x 256^2, h = 64, inverse-quadratic), times the launches with HIP events and prints, per kernel,
the share of wave cycles per phase, the cycles per K step per SIMD (768 = matrix pipe saturated by 12 MFMAs) and the
shader clock under this load (longest wave's cycles / launch time)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from histogan_amd import hist as HH  # noqa: E402
from histogan_amd._lib import check, lib  # noqa: E402

dev = torch.device('cuda:0')
B, S, h = 32, 256, 64
x = torch.rand(B, 3, S, S, device=dev)
cfg = HH.HistConfig(h=h, insz=S, method='inverse-quadratic', sigma=0.02)
p, keep = HH._make_params(x, cfg)
fb, bb = HH._ws_bytes(p)
out = torch.empty(B, 3, h, h, device=dev); sums = torch.empty(B, device=dev); gx = torch.empty_like(x)
gout = torch.rand(B, 3, h, h, device=dev) - 0.5
ws = torch.empty(max(fb, bb, 4), dtype=torch.uint8, device=dev)
cache = torch.empty(B, S * S, 8, device=dev)
p.proj_cache = cache.data_ptr()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
probe = (ctypes.c_ulonglong * 16)()
fwd = lambda: check(lib.hg_rgbuv_hist_fwd(ctypes.byref(p), x.data_ptr(), out.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws.numel(), st), 'fwd')
bwd = lambda: check(lib.hg_rgbuv_hist_bwd(ctypes.byref(p), x.data_ptr(), gout.data_ptr(), out.data_ptr(), sums.data_ptr(), gx.data_ptr(), ws.data_ptr(), ws.numel(), st), 'bwd')
for _ in range(3):
    fwd(); bwd()
torch.cuda.synchronize()
lib.hg_debug_hist_probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.hg_debug_hist_probe(probe)          # reset
N = 10
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(N):
    ev[0].record(); fwd(); ev[1].record(); bwd(); ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
lib.hg_debug_hist_probe(probe)
res = {}
for k, name, ms, total_steps in ((0, 'k_hist_fwd', tf / N, B * S * S / 2.0), (1, 'k_hist_bwd', tb / N, float(B * S * S))):
    v = [int(probe[8 * k + i]) for i in range(8)]
    waves = v[0] / N
    tot = v[1]
    res[name] = dict(launch_ms_incl_helpers=round(ms, 4), waves_per_launch=waves,
                     share=dict(prologue=v[2] / tot, projection_or_pixel_state=v[3] / tot, k_loop=v[4] / tot, epilogue=v[5] / tot),
                     wave_cycles_mean=tot / v[0], longest_wave_cycles=v[6],
                     shader_clock_mhz=v[6] / (ms * 1e3),
                     # a SIMD's two waves run their K loops side by side: 2 x (steps per wave) steps in one wave's K-loop cycles
                     k_loop_cycles_per_step_per_simd=(v[4] / v[0]) / (total_steps / waves) / 2.0,
                     all_cycles_per_step_per_simd=(tot / v[0]) / (total_steps / waves) / 2.0)
print(json.dumps(res, indent=1))
