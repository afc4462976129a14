#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s5_r06; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_flownet2.py -m gpu -x -q -k "splitk" 2>&1 | tail -3
for v in 0 1 0 1; do
  VV_FN2_SPLITK_INKERNEL=$v python tools/bench_flownet2.py > $O/fn2_inkernel$v.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/fn2_inkernel$v.json'));print('inkernel=$v', d['ms_per_pair_gpu'], d['ms_per_pair_wall'])"
done
# fp32 transposed-conv weight gradient: 64-pixel tiles + two LDS buffers vs the 128-pixel form, same box
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-secondary"
for v in 0 1 0 1; do
  VV_WGRADT_TILE64=$v $B --steps 20 --warmup 5 > $O/b256_t64_$v.json 2>/dev/null
  VV_WGRADT_TILE64=$v $B --batch 32 --steps 50 --warmup 5 > $O/b32_t64_$v.json 2>/dev/null
  python - <<PY
import json
for n in ('b256','b32'):
    d=json.load(open('$O/%s_t64_$v.json'%n)); print(n,'tile64=$v', d['value'], d['ms_per_step'], d['roofline']['wgrad']['family_ms_per_step'], d['roofline']['wgrad']['frac'])
PY
done
VV_WGRADT_TILE64=1 $B --steps 10 --no-graph --breakdown 2>&1 >/dev/null | grep -i "wgradT"
VV_WGRADT_TILE64=0 $B --steps 10 --no-graph --breakdown 2>&1 >/dev/null | grep -i "wgradT"
