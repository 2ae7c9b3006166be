#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: registers, spills, occupancy per kernel.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage -I include -c X.hip -o /dev/null 2> res.txt
    python tools/kernel_resources.py res.txt [name filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2:]
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
rows = []
for b, d in zip(blocks, dem):
    g = lambda k: int(m.group(1)) if (m := re.search(k + r': (\d+)', b)) else -1
    rows.append((d.replace('(anonymous namespace)::', '').replace('void ', ''), g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                 g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
print(f'{len(rows)} kernels, {sum(1 for r in rows if r[3] > 0)} with scratch')
for r in rows:
    if r[3] > 0 or (filt and any(f in r[0] for f in filt)):
        print(f'{r[0][:100]:100s} vgpr {r[1]:4d} agpr {r[2]:4d} scratch {r[3]:5d} occ {r[4]} lds {r[5]}')
