#!/bin/bash
# round-6 session 4: split-K finished in the conv kernel (FlowNet2): parity + A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s4_r06; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_flownet2.py tests/test_gpu_fullsize.py -m gpu -x -q -k "flownet2 or conv2d" 2>&1 | tail -5
for v in 0 1 0 1; do
  VV_FN2_SPLITK_INKERNEL=$v python tools/bench_flownet2.py > $O/fn2_inkernel$v.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/fn2_inkernel$v.json'));print('inkernel=$v', d['ms_per_pair_gpu'], d['ms_per_pair_wall'])"
done
