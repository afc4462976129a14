#!/bin/bash
# Run on the GPU box (gpurun) as the LAST thing of the round: the round-4 artefacts of profiles/, into gpurun_out/prof_r04/.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles_r04.sh'
# Counter passes never share a rocprofv3 invocation with each other or with tracing domains other than --kernel-trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r04
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
# The bench line first, on the fresh box, the way the driver runs it (after ten minutes of profiling runs the same command measured
# 1-2 % lower); its `traffic` comes from the counter files already in profiles/ -- `traffic_source` says STALE if the kernels changed
# since they were taken, in which case run this script twice.
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err
# Per-kernel counters and per-kernel durations are collected on the ONE-STREAM schedule (VV_GRAPH_OVERLAP=0: the captured train step
# without its parallel weight-gradient branch; VV_FN2_OVERLAP=0: FlowNetSD behind the FlowNetC -> S1 -> S2 chain): with two kernels
# in flight a device-wide counter or a launch duration no longer belongs to one kernel.  bench.py's roofline durations are measured
# the same way (its eager, one-stream event steps).  The default (overlapped) schedules are profiled separately below (*_overlap).
export VV_GRAPH_OVERLAP=0 VV_FN2_OVERLAP=0
pmc "" 5 $B --steps 3 --warmup 2 --no-forward-timing
pmc _bf16_full_b512 5 $C4 --steps 3 --warmup 2 --no-forward-timing
pmc _flownet2 13 $F
for t in "" _bf16_full_b512 _flownet2; do cp $O/pmc_hbm_traffic$t.json $R/profiles/r04_pmc_hbm_traffic$t.json; done   # bench.py reads these: same run
# (the counter passes run train steps only -- --no-forward-timing -- so that total bytes / Adam launches = HBM bytes per train step)
stats() {  # stats <name> <cmd...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- "$@" > $O/bench_${name}_under_rocprof.json 2> /dev/null
  cp $O/st/*/*kernel_stats.csv $O/kernel_stats_$name.csv
  cp $O/st/*/*kernel_trace.csv $O/kernel_trace_$name.csv 2>/dev/null
  rm -rf $O/st
}
stats net4_b256 $B --steps 20 --warmup 5
python $R/tools/step_gaps.py $O/kernel_trace_net4_b256.csv > $O/step_gaps_net4_b256.txt
stats bf16_full_b512 $C4 --steps 10 --warmup 3
stats net4_b32 $B --batch 32 --steps 30 --warmup 5
python $R/tools/step_gaps.py $O/kernel_trace_net4_b32.csv x > $O/step_gaps_net4_b32.txt
stats flownet2 $F
unset VV_GRAPH_OVERLAP VV_FN2_OVERLAP
stats net4_b256_overlap $B --steps 20 --warmup 5
stats net4_b32_overlap $B --batch 32 --steps 30 --warmup 5
stats flownet2_overlap python $R/tools/bench_flownet2.py
rm -f $O/kernel_trace_*.csv
python $R/tools/profile_flownet2_layers.py 2> /dev/null > $O/flownet2_layers.txt
$B --batch 32 --steps 30 --no-graph --breakdown > /dev/null 2> $O/breakdown_net4_b32.txt
$B --steps 10 --no-graph --breakdown > /dev/null 2> $O/breakdown_net4_b256.txt
$C4 --steps 10 --warmup 3 --no-graph --breakdown > /dev/null 2> $O/breakdown_bf16_full_b512.txt
# the all-bf16 3x3 family launch by launch (round-4 kernel and the round-3 kernel on the same tensors) + the steady-state loop calibration
python $R/tools/ubench_conv16.py 10 > $O/ubench_conv16.txt 2>/dev/null
python $R/tools/ubench_conv16.py 10 legacy > $O/ubench_conv16_legacy.txt 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $R/tools/ubench/gemm16_loop.hip -o /tmp/g16loop 2>/dev/null && /tmp/g16loop > $O/gemm16_loop_calibration.txt
# config 4 as its own process (the bench line's cfg4_* record runs behind the fp32 workload of the same process and reads 1 - 3 % lower),
# twice around the round-3-kernel run of the same box
$C4 --steps 20 --warmup 5 > $O/bench_bf16_full_b512.json 2>/dev/null
VV_CONV_GEMM16=0 $C4 --steps 10 --warmup 3 > $O/bench_bf16_full_b512_round3_conv_kernel.json 2>/dev/null
$C4 --steps 20 --warmup 5 >> $O/bench_bf16_full_b512.json 2>/dev/null
VV_FN2_WINO=0 python $R/tools/bench_flownet2.py > $O/bench_flownet2_no_winograd.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'], d['roofline']['traffic_source'])
for k,v in d['configs'].items(): print(k, v.get('value'), v.get('unit'), v.get('ms_per_step') or v.get('ms_per_pair'))
print(d['cpu_baseline']['sample'])
PY
