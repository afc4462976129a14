#!/bin/bash
# Run on the GPU box (gpurun): the round-2 artefacts of profiles/, into gpurun_out/prof_r02/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles_r02.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_mfma -- $B --steps 3 --warmup 2 > /dev/null 2>&1
python $R/tools/make_pmc_mfma.py $O/pmc_mfma/*/*counter_collection.csv $O/pmc_mfma_busy.json
python $R/tools/make_pmc_traffic.py $O/pmc_fetch/*/*counter_collection.csv $O/pmc_write/*/*counter_collection.csv $O/pmc_hbm_traffic.json
cp $O/pmc_hbm_traffic.json $R/profiles/r02_pmc_hbm_traffic.json      # bench.py reports `traffic` from this file: same run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fn2stats -- python $R/tools/bench_flownet2.py --eager > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -- $B --precision bf16 --model full --batch 512 --steps 10 --warmup 3 > $O/bench_bf16_full_b512_under_rocprof.json 2> /dev/null
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
cp $O/fn2stats/*/*kernel_stats.csv $O/flownet2_kernel_stats.csv
cp $O/stats16/*/*kernel_stats.csv $O/kernel_stats_bf16_full_b512.csv
rm -rf $O/stats $O/fn2stats $O/stats16 $O/pmc_fetch $O/pmc_write $O/pmc_mfma
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'])
for k,v in d['configs'].items(): print(k, v.get('value'), v.get('unit'), v.get('ms_per_step') or v.get('ms_per_pair'))
print(d['cpu_baseline']['sample'])
"
