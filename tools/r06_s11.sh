#!/bin/bash
# diagnostics on HEAD: eval per-launch table at the scoring batch, fp32 / bf16 per-launch tables
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s11_r06; rm -rf $O; mkdir -p $O
cd $R
python tools/eval_breakdown.py 2048 5 2>/dev/null | tee $O/eval_b2048.txt
python tools/eval_breakdown.py 512 10 2>/dev/null | tee $O/eval_b512.txt | tail -3
python bench.py --no-graph --breakdown --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $O/bd_fp32.json 2> $O/bd_fp32.txt; head -60 $O/bd_fp32.txt
python bench.py --no-graph --breakdown --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --batch 32 > $O/bd_b32.json 2> $O/bd_b32.txt; tail -2 $O/bd_b32.txt
