#!/bin/bash
# eval scoring rate against the scoring batch (sum of the 22 launches of the plan, HIP events)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for b in 512 1024 2048 4096 8192; do timeout 300 python tools/eval_breakdown.py $b 5 2>/dev/null | tail -1; done
