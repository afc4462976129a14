#!/bin/bash
# A/B: conv_gemm16p 64-wide N tiles with a 2 x 2 wave tile (VV_G16_W22=1) against the 4 x 1 form; parity first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s12_r06; rm -rf $O; mkdir -p $O
cd $R
VV_G16_W22=1 timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1 0 1; do echo "== W22=$v"; VV_G16_W22=$v python tools/ubench_conv16.py 20 2>/dev/null | grep -E "^(conv|dgrad)(2|3|10|11|4)|sum|step" ; done | tee $O/ubench.txt
C4="python bench.py --no-cpu-baseline --no-secondary --precision bf16 --model full --batch 512 --steps 20 --warmup 5"
for v in 0 1 0 1; do VV_G16_W22=$v $C4 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('cfg4 W22=$v', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],1))"; done | tee $O/cfg4.txt
