"""Who launches what in a replayed step / forward: first-party kernels (libvecvad_hip.so), framework kernels (at::native ...) and runtime
blits (__amd_rocclr_*), counted from a rocprofv3 --kernel-trace CSV between two launches of a marker kernel.

    python tools/launch_census.py <kernel_trace.csv> <marker substring> [periods]

Takes the last `periods` (default 8) complete periods between consecutive marker launches (the steady state: hipGraph replays) and
prints one JSON object: per class the launches per period (min / max over the periods) and the names of everything not first-party."""
import collections
import csv
import json
import sys


def cls(name):
    if name.startswith('__amd_rocclr'):
        return 'runtime'
    if 'at::native' in name or 'at::cuda' in name or name.startswith('void at::'):
        return 'framework'
    return 'first_party'


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marker = sys.argv[2]
    periods = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    spans = list(zip(idx[:-1], idx[1:]))[-periods:]
    per = []
    other = collections.Counter()
    for a, b in spans:
        c = collections.Counter()
        for r in rows[a + 1:b + 1]:
            k = cls(r['Kernel_Name'])
            c[k] += 1
            if k != 'first_party':
                other[r['Kernel_Name'][:90]] += 1
        per.append(c)
    out = {'marker': marker, 'periods': len(spans)}
    for k in ('first_party', 'framework', 'runtime'):
        v = [c[k] for c in per]
        out[k + '_launches_per_period'] = {'min': min(v), 'max': max(v)} if v else None
    out['not_first_party'] = {k: round(n / max(len(spans), 1), 2) for k, n in other.items()}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
