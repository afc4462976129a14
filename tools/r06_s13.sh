#!/bin/bash
# final check of the round: the whole GPU suite + smoke + the driver's bench command on the committed tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s13_r06; rm -rf $O; mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err
python tools/print_bench.py < $O/bench.json        # (reads stdin: without the redirect it waits until gpurun's limit)
