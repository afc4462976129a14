#!/bin/bash
# round-6 session 2: the stripped library through the whole GPU suite, the Winograd legs of the parity bisect, the new roofline records
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2_r06; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
rm -f gpurun_out/observed.jsonl
for sw in "VV_WINOGRAD=0" "VV_WINOGRAD_WGRAD=0" "VV_WINOGRAD=1"; do
  echo "{\"switch\": \"$sw\"}" >> gpurun_out/observed.jsonl
  env $sw timeout 600 python -m pytest tests/test_gpu_scripts.py -q -x -k "test_train_then_test_scripts_match_reference and fp32" 2>&1 | tail -1
done
cp gpurun_out/observed.jsonl $O/observed_switches.jsonl
cat $O/observed_switches.jsonl
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/bench_headline.json 2>$O/err1.txt
python bench.py --no-cpu-baseline --no-secondary --precision bf16 --model full --batch 512 --steps 20 --warmup 5 > $O/bench_c4.json 2>$O/err2.txt
python - <<PY
import json
for n in ('headline','c4'):
    try:
        d=json.load(open('$O/bench_%s.json'%n)); r=d['roofline']; print(n, d['value'], d['ms_per_step'], r['frac']); print(json.dumps(r.get('wgrad'))); print(json.dumps(r.get('bn_bwd')))
    except Exception as e: print(n, 'ERR', e)
PY
tail -3 $O/err1.txt
