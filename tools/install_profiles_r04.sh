#!/bin/bash
# Copy what tools/collect_profiles_r04.sh left under gpurun_out/prof_r04/ into profiles/ under the round-4 names.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r04; P=$R/profiles
cp $O/bench.json $P/r04_bench.json
cp $O/bench_net4_b256_under_rocprof.json $P/r04_bench_under_rocprof.json
cp $O/kernel_stats_net4_b256.csv $P/r04_kernel_stats.csv
cp $O/kernel_stats_flownet2.csv $P/r04_flownet2_kernel_stats.csv
cp $O/kernel_stats_flownet2_overlap.csv $P/r04_flownet2_kernel_stats_overlap.csv
for n in bf16_full_b512 net4_b32 net4_b256_overlap net4_b32_overlap; do
  cp $O/kernel_stats_$n.csv $P/r04_kernel_stats_$n.csv
  cp $O/bench_${n}_under_rocprof.json $P/r04_bench_${n}_under_rocprof.json
done
cp $O/bench_flownet2_under_rocprof.json $P/r04_bench_flownet2_under_rocprof.json
cp $O/bench_flownet2_overlap_under_rocprof.json $P/r04_bench_flownet2_overlap_under_rocprof.json
for t in "" _bf16_full_b512 _flownet2; do
  cp $O/pmc_hbm_traffic$t.json $P/r04_pmc_hbm_traffic$t.json
  cp $O/pmc_mfma_busy$t.json $P/r04_pmc_mfma_busy$t.json
done
for n in net4_b256 net4_b32; do cp $O/step_gaps_$n.txt $P/r04_step_gaps_$n.txt; done
for n in net4_b256 net4_b32 bf16_full_b512; do cp $O/breakdown_$n.txt $P/r04_breakdown_$n.txt; done
for f in ubench_conv16.txt ubench_conv16_legacy.txt gemm16_loop_calibration.txt bench_bf16_full_b512.json bench_bf16_full_b512_round3_conv_kernel.json bench_flownet2_no_winograd.json; do [ -f $O/$f ] && cp $O/$f $P/r04_$f; done
[ -f $O/flownet2_layers.txt ] && cp $O/flownet2_layers.txt $P/r04_flownet2_layers.txt
python - <<PY
import json
d = json.load(open('$P/r04_bench.json'))
print('installed; bench:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'])
PY
