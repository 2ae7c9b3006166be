#!/usr/bin/env python3
"""profiles/r<NN>_pmc_traffic.json (round given as the 5th argument, default 06): FETCH_SIZE / WRITE_SIZE per launch (rocprofv3 --kernel-trace --pmc, separate passes, KB) of
bench.py's roofline launches, stamped with the digest of the kernel source they were measured on (bench.py reports `traffic`
only while that digest still matches):
  * tools/wino_pmc.sh      <out>/wino_roofline_traffic.txt   hg_wino_conv2d / hg_wino_wgrad at 256 -> 128 channels, 64 x 64, batch 32
  * tools/conv_traffic.sh  <out>/traffic.txt                 the direct kernels of the same layer + the dense histogram kernels (configs[1])
  * tools/hist_traffic.sh  <out>/traffic.txt                 HG_HIST_METHOD=thresholding
Correction: FETCH_SIZE x 2 for EVERY read pattern.  tools/ubench/fetch_calib.hip reads 1 GiB (4 x the Infinity Cache) exactly once
in four patterns -- 16 B / lane contiguous, 4 B / lane contiguous, k_wino's overlapping patch rows, k_wino_wgrad's 64-byte segments --
and the counter answers 0.500, 0.500, 0.531, 0.560 GiB (profiles/r05_fetch_calibration.txt): the L2 fetches 128-byte lines and each
is tallied as 64 B, whatever the lane width (the guide states the factor for 16 B / lane only; round 4 took 4-byte-per-lane loads
"as reported", which under-stated the direct kernels' reads by half).  WRITE_SIZE as reported (k_wino's 67 108 864-byte output reads
65 536.0 KB).  Infinity-Cache hits are counted: this is fabric traffic, an upper bound on HBM traffic.

    python tools/make_traffic_record_rounds.py <wino txt> <conv txt> <thr txt> <commit> [round]"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r'## (.*)', line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.match(r'\s+(\w+)\s+([\d.]+)\s+\(n=', line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
        m = re.match(r'\s+duration_us\(avg under PMC\) = ([\d.]+)', line)
        if m and cur:
            out[cur]['duration_us'] = float(m.group(1))
    return out


def digest(name):
    with open(os.path.join(ROOT, 'histogan_amd', 'csrc', name), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def find(d, sub):
    for k, v in d.items():
        if sub in k and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            return k, v
    raise SystemExit(f'no kernel matching {sub!r} with both counters')


def entry(d, sub, src, commit):
    k, v = find(d, sub)
    return dict(fetch_bytes=2 * v['FETCH_SIZE'] * 1024, write_bytes=v['WRITE_SIZE'] * 1024, fetch_kb_reported=v['FETCH_SIZE'],
                duration_us_under_pmc=v.get('duration_us'), kernel=k[:90], comment='FETCH_SIZE x 2 (calibrated), WRITE_SIZE as reported',
                source=src, source_sha16=digest(src), commit=commit)


wino, conv, thr, commit = parse(sys.argv[1]), parse(sys.argv[2]), parse(sys.argv[3]), sys.argv[4]
bench = {
    'k_wino_fwd_256_128_64_b32': entry(wino, 'k_wino<2, 2, 8, false>', 'hg_wino.hip', commit),
    'k_wino_wgrad_256_128_64_b32': entry(wino, 'k_wino_wgrad(', 'hg_wino.hip', commit),
    'k_conv_fwd_256_128_64_b32': entry(conv, 'k_conv<', 'hg_conv.hip', commit),
    'k_wgrad_256_128_64_b32': entry(conv, 'k_wgrad<', 'hg_conv.hip', commit),
    'k_hist_bwd_c2': entry(conv, 'k_hist_bwd<', 'hg_hist.hip', commit),
    'k_hist_fwd_c2': entry(conv, 'k_hist_fwd<', 'hg_hist.hip', commit),
}
t, fetch, write = {}, 0.0, 0.0
for name in ('k_thr_fwd_lean', 'k_hist_finish', 'k_thr_bwd_lean'):
    e = entry(thr, name, 'hg_hist.hip', commit)
    t[name] = {k: e[k] for k in ('fetch_kb_reported', 'fetch_bytes', 'write_bytes', 'duration_us_under_pmc')}
    fetch += e['fetch_bytes']
    write += e['write_bytes']
t['total_fabric_bytes_fwd_bwd'] = fetch + write
t['algorithmic_bytes_fwd_bwd'] = 78643200
bench['thr_fwd_bwd_c2'] = dict(fetch_bytes=fetch, write_bytes=write, comment='k_thr_fwd_lean + k_hist_finish + k_thr_bwd_lean; FETCH_SIZE x 2',
                               source='hg_hist.hip', source_sha16=digest('hg_hist.hip'), commit=commit)
rec = dict(_note=' '.join(__doc__.split('Correction: ')[1].split('\n\n')[0].split()), thresholding_b32_256x256_h64=t, bench=bench)
rnd = sys.argv[5] if len(sys.argv) > 5 else '06'
with open(os.path.join(ROOT, 'profiles', f'r{rnd}_pmc_traffic.json'), 'w') as f:
    json.dump(rec, f, indent=1)
for k, v in bench.items():
    print(f'{k:30s} fetch {v["fetch_bytes"] / 1e6:8.1f} MB  write {v["write_bytes"] / 1e6:7.1f} MB')
