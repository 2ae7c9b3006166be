#!/usr/bin/env python3
"""Where does the wall time of a train step go?  Sweep-line over a rocprofv3 kernel trace (.db) of tools/step_kernels.py:
time with a matrix-pipe kernel running (k_conv / k_wgrad / k_hist_* / rocBLAS), time with ONLY memory-bound kernels
running (the part a fusion or a better overlap would remove), idle time (launch gaps / dependencies), per step.  The
window is [2nd-last .. last] x k_diffgrad launches counted from the end: `steps` whole steps of the steady state.

    python tools/trace_timeline.py trace.db STEPS [SKIP]"""
import re
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
c = sqlite3.connect(db)
rows = c.execute('select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s '
                 'on d.kernel_id = s.id order by d.start').fetchall()
MFMA = re.compile(r'k_conv|k_wgradI|k_winoI|k_wino_wgradE|k_hist_fwd|k_hist_bwd|Cijk_')
opt = [i for i, r in enumerate(rows) if 'k_diffgrad' in r[2]]
per_step = len(opt) // steps                       # optimizer launches per step (2: D and G)
first = opt[skip * per_step - 1] + 1               # after the last optimizer launch of step `skip`
last = opt[-1]
win = rows[first:last + 1]
n = steps - skip
t0, t1 = win[0][0], max(r[1] for r in win)
ev = []
for s, e, name in win:
    k = 0 if MFMA.search(name) else 1
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
cnt = [0, 0]
prev = t0
acc = {'mfma': 0, 'mfma+hbm': 0, 'hbm only': 0, 'idle': 0}
only = {}
for t, d, k in ev:
    dt = t - prev
    if dt > 0:
        if cnt[0] and cnt[1]: acc['mfma+hbm'] += dt
        elif cnt[0]: acc['mfma'] += dt
        elif cnt[1]: acc['hbm only'] += dt
        else: acc['idle'] += dt
    cnt[k] += d
    prev = t
# which memory-bound kernels run while no matrix kernel does: attribute exclusive time per kernel name
mf = sorted((s, e) for s, e, nme in win if MFMA.search(nme))
merged = []
for s, e in mf:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
import bisect
starts = [m[0] for m in merged]
def uncovered(s, e):
    tot = e - s
    i = max(bisect.bisect_right(starts, s) - 1, 0)
    while i < len(merged) and merged[i][0] < e:
        a, b = max(s, merged[i][0]), min(e, merged[i][1])
        if b > a: tot -= b - a
        i += 1
    return tot
for s, e, nme in win:
    if not MFMA.search(nme):
        short = re.sub(r'_ZN\d*_GLOBAL__N_1\d*', '', nme)
        short = re.sub(r'_ZN2at6native\d*', 'at::', short)[:70]
        o = only.setdefault(short, [0, 0, 0])
        o[0] += 1; o[1] += e - s; o[2] += uncovered(s, e)
wall = (t1 - t0) / 1e6 / n
print(f'{n} steps, wall {wall:.2f} ms/step')
for k, v in acc.items():
    print(f'  {k:9s} {v / 1e6 / n:7.2f} ms/step  {v / (t1 - t0) * 100:5.1f} %')
print('memory-bound kernels: ms/step total, ms/step NOT under a matrix kernel (sums can exceed "hbm only": they overlap each other)')
for k, (cn, tot, un) in sorted(only.items(), key=lambda kv: -kv[1][2])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print(f'  {tot / 1e6 / n:7.3f} {un / 1e6 / n:7.3f}  {cn / n:6.1f} calls  {k}')

# the largest idle gaps (nothing running) of the window: the kernel that ended before and the one that started after
spans = sorted((s_, e_, nme) for s_, e_, nme in win)
gaps = []
cur_end, cur_name = spans[0][1], spans[0][2]
for s_, e_, nme in spans[1:]:
    if s_ > cur_end:
        gaps.append((s_ - cur_end, cur_name, nme, cur_end - t0))
    if e_ > cur_end:
        cur_end, cur_name = e_, nme
def sh(nme):
    nme = re.sub(r'_ZN\d*_GLOBAL__N_1\d*', '', nme)
    return re.sub(r'_ZN2at6native\d*', 'at::', nme)[:48]
print(f'{len(gaps) / n:.0f} idle gaps per step; gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000) / n:.1f} per step, '
      f'{sum(g[0] for g in gaps if g[0] > 20000) / 1e6 / n:.2f} ms/step; <= 20 us: {sum(g[0] for g in gaps if g[0] <= 20000) / 1e6 / n:.2f} ms/step')
for g in sorted(gaps, key=lambda g: -g[0])[:24]:
    print(f'  {g[0] / 1e3:8.1f} us at +{g[3] / 1e6:8.2f} ms  after {sh(g[1])}  before {sh(g[2])}')

# the longest stretches without a matrix-pipe kernel, with what ran inside them
nomf = []
prev_end = t0
for s_, e_ in merged:
    if s_ > prev_end: nomf.append((prev_end, s_))
    prev_end = max(prev_end, e_)
print(f'{len(nomf) / n:.0f} stretches without a matrix kernel per step; the longest:')
other = sorted((s_, e_, nme) for s_, e_, nme in win if not MFMA.search(nme))
for a_, b_ in sorted(nomf, key=lambda iv: iv[0] - iv[1])[:int(sys.argv[5]) if len(sys.argv) > 5 else 16]:
    inside = [(min(e_, b_) - max(s_, a_), nme) for s_, e_, nme in other if s_ < b_ and e_ > a_]
    agg = {}
    for d_, nme in inside: agg[sh(nme)] = agg.get(sh(nme), 0) + d_
    top = ', '.join(f'{k} {v / 1e3:.0f}us' for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:4])
    print(f'  {(b_ - a_) / 1e3:7.1f} us at +{(a_ - t0) / 1e6:7.2f} ms: {top}')
