#!/bin/bash
# stream priorities of the two branches of the captured step (experiment)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
B="python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5 --no-forward-timing"
run() { echo -n "$1: "; env $2 $B 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
run "default        " "X=1"
run "main high      " "VV_MAIN_PRIO=-1"
run "side high      " "VV_SIDE_PRIO=-1"
done
for i in 1 2; do
echo -n "b32 default: "; $B --batch 32 --steps 50 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],4))"
echo -n "b32 main high: "; VV_MAIN_PRIO=-1 $B --batch 32 --steps 50 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],4))"
done
