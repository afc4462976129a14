"""profiles/r01_pmc_mfma_busy.json from a rocprofv3 counter-collection CSV of
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
MFMA-busy fraction per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), both summed over the launches."""
import collections
import csv
import json
import re
import sys


def main():
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        k = re.sub(r'^void ', '', r['Kernel_Name']).replace('(anonymous namespace)::', '')
        k = k.split('(')[0] if '<' not in k.split('(')[0] else k[:k.index('>') + 1]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_BUSY_CU_CYCLES':
            n[k] += 1
    rows = []
    for k, v in tot.items():
        busy, cu = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), v.get('SQ_BUSY_CU_CYCLES', 0.0)
        if cu > 0 and busy > 0:
            rows.append({'kernel': k, 'launches': n[k], 'mfma_busy_cycles': busy, 'busy_cu_cycles': cu,
                         'mfma_busy_frac': round(busy / (4.0 * cu), 4)})
    rows.sort(key=lambda r: -r['busy_cu_cycles'])
    out = {'source': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline',
           'definition': 'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): share of the SIMD-cycles of busy CUs in which '
                         'the matrix pipe was executing (the chip clocks down to ~1.9-2.0 GHz under dense fp32 MFMA, so this is utilisation '
                         'at the sustained clock, not a fraction of the 157.3 TFLOP/s data-sheet peak)',
           'kernels': rows}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    for r in rows[:12]:
        print('%-50s %.3f' % (r['kernel'][:50], r['mfma_busy_frac']))


if __name__ == '__main__':
    main()
