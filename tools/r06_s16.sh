#!/bin/bash
# same-box A/B: PLANE == 4 mod 16 (product) against the round-4 layout (exp_libs/lib_plane8.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in A B A B; do
  if [ $v = B ]; then export VV_LIB_PATH=$R/exp_libs/lib_plane8.so; else unset VV_LIB_PATH; fi
  echo "== $v (${VV_LIB_PATH:-product: plane4})"
  python tools/ubench_conv16.py 20 2>/dev/null | grep -E "weighted"
  python bench.py --no-cpu-baseline --no-secondary --precision bf16 --model full --batch 512 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('cfg4', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],1))"
done
