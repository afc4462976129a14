#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s8_r06; rm -rf $O; mkdir -p $O
cd $R
for b in 512 1024 2048; do python tools/eval_breakdown.py $b 10 > $O/eval_b$b.txt 2>/dev/null; echo "== B=$b"; cat $O/eval_b$b.txt; done
for m in 0 1 all; do echo "== VV_WINO44_EVAL=$m B=512"; VV_WINO44_EVAL=$m python tools/eval_breakdown.py 512 10 2>/dev/null | tail -1; done
