#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 csv passes (tools/pmc_passes.sh).

    python tools/pmc_summary.py gpurun_out/pmc1 [kernel-substring ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2:] or ['k_hist_fwd', 'k_hist_bwd', 'k_hist_reduce']
    res = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, '*', '*counter_collection.csv')):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r['Kernel_Name']
                res[k][r['Counter_Name']].append(float(r['Counter_Value']))
                if 'Start_Timestamp' in r and r.get('End_Timestamp'):
                    dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    for k in sorted(res):
        if not any(w in k for w in want):
            continue
        print(f'## {k[:100]}')
        if dur[k]:
            print(f'  duration_us(avg under PMC) = {sum(dur[k])/len(dur[k]):.1f}')
        for c in sorted(res[k]):
            v = res[k][c]
            print(f'  {c:34s} {sum(v)/len(v):16.1f}   (n={len(v)})')


if __name__ == '__main__':
    main()
