#!/bin/bash
# round-6 session 1: per-switch parity margins of the script tests + same-box baselines of the unchanged library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1_r06; rm -rf $O; mkdir -p $O
cd $R
rm -f gpurun_out/observed.jsonl
for w44 in 0 dgrad; do for fbs in 0 1; do
  echo "{\"switch\": \"VV_WINO44=$w44 VV_FUSE_BN_SUMS=$fbs\"}" >> gpurun_out/observed.jsonl
  VV_WINO44=$w44 VV_FUSE_BN_SUMS=$fbs timeout 600 python -m pytest tests/test_gpu_scripts.py -q -x -k "test_train_then_test_scripts_match_reference" 2>&1 | tail -3
done; done
cp gpurun_out/observed.jsonl $O/observed_switches.jsonl
B="python bench.py --no-cpu-baseline --no-secondary"
$B --steps 20 --warmup 5 > $O/bench_headline.json 2>$O/err1.txt
$B --precision bf16 --model full --batch 512 --steps 20 --warmup 5 > $O/bench_c4.json 2>$O/err2.txt
$B --batch 32 --steps 50 --warmup 5 > $O/bench_b32.json 2>$O/err3.txt
python - <<PY
import json
for n in ('headline','c4','b32'):
    try:
        d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['execution']['launches_per_step'])
    except Exception as e: print(n, 'ERR', e)
PY
cat $O/observed_switches.jsonl
