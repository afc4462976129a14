#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s10_r06; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_flownet2.py tests/test_gpu_fullsize.py tests/test_gpu_scripts.py -m gpu -x -q -k "flownet2 or conv2d or optical" 2>&1 | tail -3
for i in 1 2; do python tools/bench_flownet2.py 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('fn2', d['ms_per_pair_gpu'], d['ms_per_pair_wall'])"; done
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
r = bench.run_flownet2(torch.device('cuda', 0))
print('ms_per_pair', r['ms_per_pair'], 'x4', r['four_pairs_per_launch']['ms_per_pair'])
for k, v in sorted(r['families_eager'].items(), key=lambda kv: -kv[1]['ms_per_forward']): print(k, round(v['ms_per_forward'], 3), round(v['tflops_executed'] or 0, 1), v['launches_per_forward'])
PY
