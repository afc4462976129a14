#!/bin/bash
# Run on the GPU box (gpurun) ONCE, on the final library of the round: the round-6 artefacts of profiles/, into gpurun_out/prof_r06/.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles_r06.sh'
# Counter passes never share a rocprofv3 invocation with each other or with tracing domains other than --kernel-trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
C4="$B --precision bf16 --model full --batch 512"
F="python $R/tools/bench_flownet2.py --eager"
pmc() {  # pmc <tag> <runs> <cmd...>: FETCH_SIZE / WRITE_SIZE passes -> pmc_hbm_traffic<tag>.json, MFMA-busy pass -> pmc_mfma_busy<tag>.json
  local tag=$1 runs=$2; shift 2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- "$@" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pm -- "$@" > /dev/null 2>&1
  python $R/tools/make_pmc_traffic.py $O/pf/*/*counter_collection.csv $O/pw/*/*counter_collection.csv $O/pmc_hbm_traffic$tag.json $runs > $O/traffic$tag.txt
  python $R/tools/make_pmc_mfma.py $O/pm/*/*counter_collection.csv $O/pmc_mfma_busy$tag.json > $O/mfma$tag.txt
  rm -rf $O/pf $O/pw $O/pm
}
# Per-kernel counters and durations on the ONE-STREAM schedules (a device-wide counter / a launch duration belongs to one kernel only
# when nothing else is in flight); bench.py's roofline durations are measured the same way (its eager, one-stream event steps).
export VV_GRAPH_OVERLAP=0 VV_FN2_OVERLAP=0
pmc "" 5 $B --steps 3 --warmup 2 --no-forward-timing
pmc _bf16_full_b512 5 $C4 --steps 3 --warmup 2 --no-forward-timing
pmc _flownet2 13 $F
for t in "" _bf16_full_b512 _flownet2; do cp $O/pmc_hbm_traffic$t.json $R/profiles/r06_pmc_hbm_traffic$t.json; done   # bench.py reads these: same run
stats() {  # stats <name> <cmd...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- "$@" > $O/bench_${name}_under_rocprof.json 2> /dev/null
  cp $O/st/*/*kernel_stats.csv $O/kernel_stats_$name.csv
  cp $O/st/*/*kernel_trace.csv $O/kernel_trace_$name.csv 2>/dev/null
  rm -rf $O/st
}
stats net4_b256 $B --steps 20 --warmup 5
stats bf16_full_b512 $C4 --steps 10 --warmup 3
stats net4_b32 $B --batch 32 --steps 30 --warmup 5
stats flownet2 $F
unset VV_GRAPH_OVERLAP VV_FN2_OVERLAP
# the default (overlapped, graph-replayed) schedules: launch census of a replayed step / forward
stats net4_b256_overlap $B --steps 20 --warmup 5 --no-forward-timing
python $R/tools/idle_gaps.py $O/kernel_trace_net4_b256_overlap.csv adam_bucketed_kernel 8 3 > $O/idle_gaps_net4_b256.txt
stats flownet2_overlap python $R/tools/bench_flownet2.py
python - <<PY > $O/launch_census.json
import json, subprocess, sys
out = {}
for name, trace, marker in (('net4_b256_train_step', '$O/kernel_trace_net4_b256_overlap.csv', 'adam_bucketed_kernel'),
                            ('flownet2_forward', '$O/kernel_trace_flownet2_overlap.csv', 'prep_sum_kernel')):
    try:
        out[name] = json.loads(subprocess.check_output([sys.executable, '$R/tools/launch_census.py', trace, marker]).decode())
    except Exception as e:
        out[name] = {'error': repr(e)}
sys.path.insert(0, '$R')
from vec_vad_amd import build as B
out['library_build'] = B.wanted()[1][:16]
print(json.dumps(out, indent=1))
PY
rm -f $O/kernel_trace_*.csv
$B --batch 32 --steps 30 --no-graph --breakdown > /dev/null 2> $O/breakdown_net4_b32.txt
$B --steps 10 --no-graph --breakdown > /dev/null 2> $O/breakdown_net4_b256.txt
$C4 --steps 10 --warmup 3 --no-graph --breakdown > /dev/null 2> $O/breakdown_bf16_full_b512.txt
python $R/tools/eval_breakdown.py 2048 10 > $O/breakdown_eval_net4_b2048.txt 2>/dev/null
# the bench line last, the way the driver runs it; its `traffic` now comes from the counter files of THIS library
cp $O/launch_census.json $R/profiles/r06_launch_census.json
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
$B --steps 20 --warmup 5 > $O/bench_headline_only.json 2>/dev/null
$C4 --steps 20 --warmup 5 > $O/bench_bf16_full_b512.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'], d['roofline']['traffic_source'])
for k,v in d['configs'].items(): print(k, v.get('value'), v.get('unit'), v.get('ms_per_step') or v.get('ms_per_pair'))
print(d['cpu_baseline']['sample'])
PY
