#!/bin/bash
# same-box A/B of a variant library (exp_libs/$1) against the product: parity of the bf16 kernels on the variant, ubench, config-4 step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
V=$R/exp_libs/$1
VV_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -2
for v in A B A B; do
  if [ $v = B ]; then export VV_LIB_PATH=$V; else unset VV_LIB_PATH; fi
  echo "== $v (${VV_LIB_PATH:-product})"
  UB_H=${UB_H:-16} python tools/ubench_conv16.py 20 2>/dev/null | grep -E "H=|weighted" | cut -c1-60
  python bench.py --no-cpu-baseline --no-secondary --precision bf16 --model full --batch 512 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('cfg4', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],1))"
done
