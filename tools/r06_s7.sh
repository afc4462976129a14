#!/bin/bash
# round-6 session 7: BatchNorm-backward sums in the transposed conv's data gradient (fp32 + all-bf16): parity, then A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s7_r06; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bf16.py tests/test_gpu_fullsize.py tests/test_gpu_scripts.py tests/test_gpu_longtrain.py -m gpu -x -q 2>&1 | tail -6
B="python bench.py --no-cpu-baseline --no-secondary"
for v in 0 1 0 1; do
  VV_FUSE_BN_SUMS_T=$v $B --steps 20 --warmup 5 > $O/b256_$v.json 2>/dev/null
  VV_FUSE_BN_SUMS_T=$v $B --batch 32 --steps 50 --warmup 5 > $O/b32_$v.json 2>/dev/null
  VV_FUSE_BN_SUMS_T=$v $B --precision bf16 --model full --batch 512 --steps 20 --warmup 5 > $O/c4_$v.json 2>/dev/null
  python - <<PY
import json
for n in ('b256','b32','c4'):
    d=json.load(open('$O/%s_$v.json'%n)); print(n,'fuseT=$v', round(d['value'],1), round(d['ms_per_step'],4), d['execution']['launches_per_step'], round(d['roofline']['bn_bwd']['family_ms_per_step'],4))
PY
done
VV_FUSE_BN_SUMS_T=1 $B --steps 10 --no-graph --breakdown 2>&1 >/dev/null | grep -i "dgradT\|bn_bwd_reduce"
VV_FUSE_BN_SUMS_T=0 $B --steps 10 --no-graph --breakdown 2>&1 >/dev/null | grep -i "dgradT\|bn_bwd_reduce"
VV_FUSE_BN_SUMS_T=1 $B --precision bf16 --model full --batch 512 --steps 10 --warmup 3 --no-graph --breakdown 2>&1 >/dev/null | grep -i "dgradT\|bn_bwd_reduce"
VV_FUSE_BN_SUMS_T=0 $B --precision bf16 --model full --batch 512 --steps 10 --warmup 3 --no-graph --breakdown 2>&1 >/dev/null | grep -i "dgradT\|bn_bwd_reduce"
