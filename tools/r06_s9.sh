#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s9_r06; rm -rf $O; mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'launches', d['execution']['launches_per_step'])
c=d['config']
print({k:c[k] for k in c if k.startswith('cfg4') or k.startswith('net4_b') or k.startswith('flownet2') or k.startswith('eval')})
print('wgrad', {k:d['roofline']['wgrad'][k] for k in ('frac','family_ms_per_step','avg_launch_us')})
print('bn_bwd', {k:d['roofline']['bn_bwd'][k] for k in ('frac','family_ms_per_step','avg_launch_us')})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
