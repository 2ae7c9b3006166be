#!/usr/bin/env python3
"""Long library launches (aten / rocBLAS / copies) of ONE plain train step in a rocprofv3 kernel trace (.db) of
tools/step_kernels.py, in launch order with the custom kernel before each -- where glue passes over big tensors hide.

    python tools/trace_long_aten.py trace.db [MIN_US=15]"""
import re
import sqlite3
import sys

db = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
c = sqlite3.connect(db)
rows = c.execute('select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s '
                 'on d.kernel_id = s.id order by d.start').fetchall()
opt = [i for i, r in enumerate(rows) if 'k_diffgrad' in r[2]]
first, last = opt[-3] + 1, opt[-1]            # the last whole step: after the G optimizer of the step before
OURS = re.compile(r'_GLOBAL__N_1\d+k_|k_conv|k_wgrad|k_hist|k_dnl|k_mod|k_up2|k_pack|k_diffgrad|k_lrelu|k_channel|k_plane|k_splitk|k_demod')
short = lambda n: re.sub(r'^_ZN2at6native', 'at::', n)[:110]
prev = ''
tot = 0.0
t0 = rows[first][0]
for s, e, name in rows[first:last + 1]:
    if OURS.search(name):
        prev = name
        continue
    us = (e - s) / 1e3
    if us >= min_us:
        tot += us
        print(f'+{(s - t0) / 1e6:7.2f} ms {us:7.1f} us  {short(name)}   <- after {re.sub(r"^_ZN12_GLOBAL__N_1[0-9]+", "", prev)[:40]}')
print(f'total {tot / 1e3:.2f} ms in library launches >= {min_us} us; step {(rows[last][1] - t0) / 1e6:.1f} ms')
