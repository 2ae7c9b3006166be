#!/usr/bin/env python3
"""One steady-state train step as a kernel SEQUENCE, from a rocprofv3 csv trace of tools/step_kernels.py
(--kernel-trace --hip-runtime-trace --output-format csv):

    rel start [us]  duration [us]  gap [us] (nothing running before this kernel)  lag [us]  queue  grid  name

`lag` = kernel start - start of the host's launch call (correlation id): a lag of a few us with a gap in front of the
kernel means the GPU waited for the HOST there (launch-bound stretch); a large lag means the launch sat in the queue and
the gap, if any, is a dependency (another stream, a barrier packet).

    python tools/step_sequence.py <trace dir> [step from the end, default 2] > sequence.txt"""
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def rows(pattern):
    for f in glob.glob(os.path.join(root, '**', pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


api = {}
for r in rows('*hip_api_trace.csv'):
    if 'aunch' in r['Function']:
        api[r['Correlation_Id']] = (int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Thread_Id', '?'))
kern = []
for r in rows('*kernel_trace.csv'):
    kern.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'),
                 r.get('Correlation_Id'), int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0),
                 int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)))
kern.sort()
opt = [i for i, k in enumerate(kern) if 'k_diffgrad' in k[2]]
# a step = from after the G optimizer launch of step n-1 to the G optimizer launch of step n (2 k_diffgrad per step)
end = opt[-1 - 2 * (back - 1)]
start = opt[-1 - 2 * back] + 1
win = kern[start:end + 1]
t0 = win[0][0]


def sh(n):
    n = re.sub(r'\.kd$', '', n)
    n = re.sub(r'_ZN\d*_GLOBAL__N_1\d*', '', n)
    n = re.sub(r'_ZN2at6native\d*', 'at::', n)
    n = re.sub(r'^_Z\d+', '', n)
    return n[:64]


MFMA = re.compile(r'k_conv|k_wgradI|k_winoI|k_wino_wgradE|k_hist_fwd|k_hist_bwd|Cijk_|k_glin')
queues = sorted({k[3] for k in win})
print(f'# step window {(win[-1][1] - t0) / 1e3:.1f} us, {len(win)} kernels, queues {queues}')
print('#   start      dur      gap      lag  host_gap q  thr   wgs  name      (M = matrix kernel; host_gap = launch call start - previous launch call start)')
cur_end = t0
busy = 0
prev_api = None
hostbound = 0.0
dep = 0.0
for s, e, name, q, cid, gx, wx in win:
    gap = max(0, s - cur_end)
    a = api.get(cid)
    lag = (s - a[0]) / 1e3 if a else float('nan')
    hg = (a[0] - prev_api) / 1e3 if (a and prev_api is not None) else float('nan')
    if a:
        prev_api = a[0]
    if gap > 0:
        if a and lag < 25:
            hostbound += gap
        else:
            dep += gap
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f} {lag:8.1f} {hg:8.1f} {queues.index(q)} '
          f'{(a[2][-3:] if a else "?"):>4} {gx // max(wx, 1):6d}  {"M " if MFMA.search(name) else "  "}{sh(name)}')
    cur_end = max(cur_end, e)
print(f'# idle in front of kernels whose launch call was < 25 us old (host-bound): {hostbound / 1e3:.1f} us; '
      f'other idle (dependencies / queued): {dep / 1e3:.1f} us')
