#!/usr/bin/env python3
"""Where the host and the GPU of each traced process spent a run: from a rocprofv3 --kernel-trace --hip-runtime-trace
--memory-copy-trace directory (csv), per process id: wall span, GPU busy time (union of kernel intervals), the HIP API calls
with the largest total time (count, total, longest), the memory copies, and the longest GPU idle gaps with the kernels on
either side.      python tools/hip_api_summary.py <dir> [top]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def rows(pattern):
    for f in glob.glob(os.path.join(root, '**', pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r


api = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
span = {}
for f, r in rows('*hip_api_trace.csv'):
    pid = r.get('Process_Id') or os.path.basename(f).split('_')[0]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    a = api[pid][r['Function']]
    a[0] += 1
    a[1] += (e - s) / 1e6
    a[2] = max(a[2], (e - s) / 1e6)
    lo, hi = span.get(pid, (s, e))
    span[pid] = (min(lo, s), max(hi, e))
kern = defaultdict(list)
for f, r in rows('*kernel_trace.csv'):
    pid = r.get('Process_Id') or os.path.basename(f).split('_')[0]
    kern[pid].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Queue_Id', '?')))
cp = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0]))
for f, r in rows('*memory_copy_trace.csv'):
    pid = r.get('Process_Id') or os.path.basename(f).split('_')[0]
    c = cp[pid][r.get('Direction', '?')]
    c[0] += 1
    c[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    c[2] += int(r.get('Bytes', 0) or 0)
for pid in sorted(set(api) | set(kern)):
    print(f'== process {pid}')
    if pid in span:
        print(f'   host API span {(span[pid][1] - span[pid][0]) / 1e6:.1f} ms')
    ks = sorted(kern.get(pid, []))
    if ks:
        busy, cur_s, cur_e = 0, ks[0][0], ks[0][1]
        gaps = []
        prev = ks[0]
        for k in ks[1:]:
            if k[0] > cur_e:
                gaps.append((k[0] - cur_e, prev[2], k[2], (cur_e - ks[0][0]) / 1e6))
                busy += cur_e - cur_s
                cur_s, cur_e = k[0], k[1]
            else:
                cur_e = max(cur_e, k[1])
            if k[1] >= cur_e:
                prev = k
        busy += cur_e - cur_s
        print(f'   kernels {len(ks)}, queues {len({k[3] for k in ks})}, GPU span {(ks[-1][1] - ks[0][0]) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms')
        for g in sorted(gaps, reverse=True)[:top]:
            print(f'      idle {g[0] / 1e6:9.2f} ms at +{g[3]:9.1f} ms  after {g[1][:44]:44s} before {g[2][:44]}')
    for fn, (n, tot, mx) in sorted(api.get(pid, {}).items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'   {fn:36s} calls {n:7d}  total {tot:10.1f} ms  longest {mx:9.2f} ms')
    for d, (n, tot, b) in cp.get(pid, {}).items():
        print(f'   copy {d:24s} n {n:6d}  total {tot:9.1f} ms  {b / 1e6:10.1f} MB')
