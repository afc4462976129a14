#!/bin/bash
# conv_gemm16p: four-plane LDS layout of a 32-channel chunk off its bank-group collision (PLANE == 4 mod 16) -- parity, ubench, counters
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -2
python tools/ubench_conv16.py 20 2>/dev/null | grep -E "H=16|H= 8|H= 4|weighted"
UB_H=16 bash tools/pmc_generic.sh pmc_g16b tools/ubench_conv16.py 5 > /dev/null 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/pmc_g16b/summary.json'))
for k,v in d.items():
    if 'gemm16' in k: print(k[:80], 'conflict', round(v['lds_conflict_frac'],3), 'lds_active', round(v['lds_active_per_cu_cycle'],3), 'mfma', round(v['mfma_busy_frac'],3))
PY
