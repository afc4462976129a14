#!/bin/bash
# Copy what tools/collect_profiles_r06.sh left under gpurun_out/prof_r06/ into profiles/ under the round-6 names.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r06; P=$R/profiles
cp $O/bench.json $P/r06_bench.json
cp $O/kernel_stats_net4_b256.csv $P/r06_kernel_stats.csv
cp $O/bench_net4_b256_under_rocprof.json $P/r06_bench_under_rocprof.json
for n in bf16_full_b512 net4_b32 net4_b256_overlap flownet2 flownet2_overlap; do
  cp $O/kernel_stats_$n.csv $P/r06_kernel_stats_$n.csv
  cp $O/bench_${n}_under_rocprof.json $P/r06_bench_${n}_under_rocprof.json
done
for t in "" _bf16_full_b512 _flownet2; do
  cp $O/pmc_hbm_traffic$t.json $P/r06_pmc_hbm_traffic$t.json
  cp $O/pmc_mfma_busy$t.json $P/r06_pmc_mfma_busy$t.json
done
cp $O/idle_gaps_net4_b256.txt $P/r06_idle_gaps_net4_b256.txt
cp $O/launch_census.json $P/r06_launch_census.json
for n in net4_b256 net4_b32 bf16_full_b512; do cp $O/breakdown_$n.txt $P/r06_breakdown_$n.txt; done
for f in breakdown_eval_net4_b2048.txt bench_headline_only.json bench_bf16_full_b512.json; do [ -f $O/$f ] && cp $O/$f $P/r06_$f; done
python - <<PY
import json
d = json.load(open('$P/r06_bench.json'))
print('installed; bench:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'])
PY
