#!/usr/bin/env python3
"""profiles/r05_pmc_traffic.json = round 4's record (histogram / thresholding / direct-convolution launches, still valid for
their unchanged sources: bench.py checks the source digest) + the Winograd roofline launch of round 5 from
tools/wino_pmc.sh's per-launch FETCH_SIZE / WRITE_SIZE passes (<out>/wino_roofline_traffic.txt: hg_wino_conv2d and
hg_wino_wgrad at 256 -> 128 channels, 64 x 64, batch 32).  k_wino / k_wino_wgrad read through 16-byte-per-lane loads
(patch rows, weight operands): FETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of the
bytes of such streams); WRITE_SIZE as reported.

    python tools/make_traffic_record_r05.py <wino_roofline_traffic.txt> <commit>"""
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
        if sub in k:
            return k, v
    raise SystemExit(f'no kernel matching {sub!r}')


pmc, commit = parse(sys.argv[1]), sys.argv[2]
with open(os.path.join(ROOT, 'profiles', 'r04_pmc_traffic.json')) as f:
    rec = json.load(f)
rec['_note_r05'] = __doc__.split('\n\n')[0].replace('\n', ' ')
for key, sub in (('k_wino_fwd_256_128_64_b32', 'k_wino<2, 2, 8, false>'), ('k_wino_wgrad_256_128_64_b32', 'k_wino_wgrad(')):
    k, v = find(pmc, sub)
    rec['bench'][key] = dict(fetch_bytes=2 * v['FETCH_SIZE'] * 1024, write_bytes=v['WRITE_SIZE'] * 1024, fetch_kb_reported=v['FETCH_SIZE'],
                             duration_us_under_pmc=v.get('duration_us'), kernel=k[:90],
                             comment='16-byte-per-lane loads: reported FETCH_SIZE x 2 (gfx950 correction); Infinity-Cache hits are counted',
                             source='hg_wino.hip', source_sha16=digest('hg_wino.hip'), commit=commit)
with open(os.path.join(ROOT, 'profiles', 'r05_pmc_traffic.json'), 'w') as f:
    json.dump(rec, f, indent=1)
print(json.dumps({k: rec['bench'][k] for k in rec['bench'] if 'wino' in k}, indent=1))
