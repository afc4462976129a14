"""Idle time between kernels of a training step, from a rocprofv3 --kernel-trace CSV (kernel_trace.csv).

    python tools/step_gaps.py <kernel_trace.csv>

Takes the last 10 occurrences of the optimiser kernel (`adam_bucketed_kernel`) as step boundaries and prints, per step: wall time between two optimizer launches,
the sum of kernel durations inside, and their difference (= launch gaps + host stalls)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'adam_bucketed_kernel' in r['Kernel_Name'] or 'adam_kernel' in r['Kernel_Name']]
for a, b in list(zip(ad[:-1], ad[1:]))[-10:]:
    seg = rows[a + 1:b + 1]
    wall = int(seg[-1]['End_Timestamp']) - int(rows[a]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    gaps = sorted(((int(y['Start_Timestamp']) - int(x['End_Timestamp'])) for x, y in zip([rows[a]] + seg[:-1], seg)), reverse=True)
    print('kernels %4d  wall %8.1f us  busy %8.1f us  idle %7.1f us  largest gaps %s' % (
        len(seg), wall / 1e3, busy / 1e3, (wall - busy) / 1e3, [round(g / 1e3, 1) for g in gaps[:5]]))
if len(sys.argv) > 2:       # any second argument: the kernel sequence of the last step, one line per launch
    a, b = ad[-2], ad[-1]
    for r in rows[a + 1:b + 1]:
        print('%8.1f us  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:110]))
