#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel trace) as a per-kernel stats table (like --stats CSV).

    python tools/rocpd_stats.py gpurun_out/prof1/r1_results.db [out.md]
"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute(
        'select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), '
        'max(d.end-d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), '
        'max(d.group_segment_size), max(d.grid_size_x*d.grid_size_y*d.grid_size_z/(d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z)) '
        'from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id '
        'group by s.kernel_name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | workgroups |',
             '|---|---|---|---|---|---|---|---|---|---|---|---|']
    for r in rows:
        name = r[0]
        if len(name) > 90:
            name = name[:87] + '...'
        lines.append(f'| `{name}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | '
                     f'{100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |')
    txt = '\n'.join(lines)
    print(txt)
    if len(sys.argv) > 2:
        with open(sys.argv[2], 'w') as f:
            f.write(txt + '\n')


if __name__ == '__main__':
    main()
