"""True device idle time inside replayed steps, from a rocprofv3 --kernel-trace CSV: union of the busy intervals of ALL kernels
(two-stream schedules overlap), idle intervals listed with the kernel that ended before and the one that starts after.

    python tools/idle_gaps.py <kernel_trace.csv> <marker substring> [min_us=10] [periods=3]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
periods = int(sys.argv[4]) if len(sys.argv) > 4 else 3
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
for a, b in list(zip(idx[:-1], idx[1:]))[-periods:]:
    seg = rows[a:b + 1]
    t_end = int(seg[0]['End_Timestamp'])
    last = seg[0]
    idle = 0
    lines = []
    for r in seg[1:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > t_end:
            idle += s - t_end
            if (s - t_end) / 1e3 >= min_us:
                lines.append('   idle %7.1f us  after %-70s before %s' % ((s - t_end) / 1e3, short(last['Kernel_Name']), short(r['Kernel_Name'])))
        if e > t_end:
            t_end, last = e, r
    wall = int(seg[-1]['End_Timestamp']) - int(seg[0]['End_Timestamp'])
    print('period: %d kernels, wall %.1f us, device idle %.1f us (%.1f %%)' % (len(seg) - 1, wall / 1e3, idle / 1e3, 100.0 * idle / wall))
    for l in lines:
        print(l)
