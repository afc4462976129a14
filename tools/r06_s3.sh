#!/bin/bash
# round-6 session 3: device idle time inside hipGraph-replayed steps (where are the gaps?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s3_r06; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-forward-timing"
tr() { # tr <name> <marker> <cmd...>
  local name=$1 marker=$2; shift 2
  rocprofv3 --kernel-trace --output-format csv -d $O/t -- "$@" > $O/bench_$name.json 2>/dev/null
  python $R/tools/idle_gaps.py $O/t/*/*kernel_trace.csv $marker 8 3 > $O/idle_$name.txt
  rm -rf $O/t
  cat $O/idle_$name.txt
}
tr net4_b256 adam_bucketed_kernel $B --steps 12 --warmup 5
tr net4_b32 adam_bucketed_kernel $B --batch 32 --steps 30 --warmup 5
tr c4 adam_bucketed_kernel $B --precision bf16 --model full --batch 512 --steps 10 --warmup 4
tr flownet2 prep_sum_kernel python $R/tools/bench_flownet2.py
