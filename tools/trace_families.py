#!/usr/bin/env python3
"""Per-kernel ms/step of a rocprofv3 kernel trace (.db) of tools/step_kernels.py; optional second trace to diff against.

    python tools/trace_families.py a.db STEPS [b.db]"""
import re
import sqlite3
import sys


def load(path, steps):
    c = sqlite3.connect(path)
    rows = c.execute('select s.kernel_name, count(*), sum(d.end-d.start) from rocpd_kernel_dispatch d join '
                     'rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name').fetchall()
    return {n: (cnt / steps, t / 1e6 / steps) for n, cnt, t in rows}


def short(n):
    n = re.sub(r'_ZN\d*_GLOBAL__N_1\d*', '', n)
    n = re.sub(r'_ZN2at6native\d*', 'at::', n)
    return n[:100]


steps = float(sys.argv[2])
a = load(sys.argv[1], steps)
b = load(sys.argv[3], steps) if len(sys.argv) > 3 and sys.argv[3] not in ('', '-') else {}
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f'total ms/step: {ta:.2f}' + (f' vs {tb:.2f} (diff {ta-tb:.2f})' if b else ''))
for k in keys[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    ca, ma = a.get(k, (0, 0))
    cb, mb = b.get(k, (0, 0))
    print(f'{ma:8.3f} ms {ca:6.1f} calls' + (f' | {mb:8.3f} ms {cb:6.1f} calls | diff {ma-mb:7.3f}' if b else '') + f'  {short(k)}')
