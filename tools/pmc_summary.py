"""Per-kernel sums of every counter in one or more rocprofv3 counter_collection.csv files (separate --pmc passes).
    python tools/pmc_summary.py out.json pass1.csv [pass2.csv ...]      prints a table, writes JSON"""
import collections
import csv
import json
import re
import sys


def short(name):
    k = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    head = k.split('(')[0]
    return head if '<' not in head else k[:k.index('>') + 1]


def main():
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for path in sys.argv[2:]:
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            launches[(k, path)].add(r['Dispatch_Id'])
    out = {}
    for k, v in tot.items():
        n = max(len(s) for (kk, _), s in launches.items() if kk == k)
        d = dict(v)
        d['launches'] = n
        wc = v.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU',
                      'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC'):
                if c in v:
                    d[c + '/WAVE_CYCLES'] = round(v[c] / wc, 4)
        if v.get('SQ_BUSY_CU_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
            d['mfma_busy_frac'] = round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * v['SQ_BUSY_CU_CYCLES']), 4)
        if v.get('SQ_LDS_IDX_ACTIVE') and 'SQ_LDS_BANK_CONFLICT' in v:
            d['lds_conflict_frac'] = round(v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE'], 4)
        if v.get('SQ_BUSY_CU_CYCLES') and 'SQ_LDS_IDX_ACTIVE' in v:
            d['lds_active_per_cu_cycle'] = round(v['SQ_LDS_IDX_ACTIVE'] / v['SQ_BUSY_CU_CYCLES'], 4)
        out[k] = d
    json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        print(k[:70])
        print('   ' + '  '.join('%s=%s' % (a, b) for a, b in sorted(d.items()) if '/' in a or a.endswith('frac') or a.endswith('cycle') or a == 'launches'))


if __name__ == '__main__':
    main()
