#!/usr/bin/env python3
"""Print the rows of a rocprofv3 kernel_stats.csv (found under a directory) with short kernel names.
    python tools/kstats.py <dir> [substring]"""
import csv, glob, os, re, sys
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
for f in glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        n = re.sub(r'\(.*', '', n).replace('void ', '')
        if want in n:
            print(f"{n[:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}  {float(r['Percentage']):5.1f} %")
