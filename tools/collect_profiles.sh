#!/bin/bash
# Run on the GPU box (gpurun): every artefact that profiles/ holds, into gpurun_out/prof_r01/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_mfma -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/make_pmc_mfma.py $O/pmc_mfma/*/*counter_collection.csv $O/pmc_mfma_busy.json
python $R/tools/make_pmc_traffic.py $O/pmc_fetch/*/*counter_collection.csv $O/pmc_write/*/*counter_collection.csv $O/pmc_hbm_traffic.json
cp $O/pmc_hbm_traffic.json $R/profiles/r01_pmc_hbm_traffic.json      # bench.py reports `traffic` from this file: same run
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python $R/tools/bench_flownet2.py > $O/flownet2_1024x448.json 2> /dev/null
python $R/tools/bench_flownet2.py 384 512 > $O/flownet2_512x384.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fn2stats -- python $R/tools/bench_flownet2.py --eager > /dev/null 2>&1
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --model full --batch 512 > $O/bench_full_b512.json 2> /dev/null
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 1024 > $O/bench_net4_b1024.json 2> /dev/null
# mixed precision (BASELINE config 4's arithmetic): the config-2 workload, config 4's own workload (Full bank, B = 512), kernel stats, HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -- python $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch16 -- python $R/bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write16 -- python $R/bench.py --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/make_pmc_traffic.py $O/pmc_fetch16/*/*counter_collection.csv $O/pmc_write16/*/*counter_collection.csv $O/pmc_hbm_traffic_bf16.json
cp $O/pmc_hbm_traffic_bf16.json $R/profiles/r01_pmc_hbm_traffic_bf16.json
python $R/bench.py --precision bf16 --steps 20 --warmup 5 > $O/bench_bf16.json 2> /dev/null
python $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --model full --batch 512 > $O/bench_bf16_full_b512.json 2> /dev/null
cp $O/stats16/*/*kernel_stats.csv $O/kernel_stats_bf16.csv
rm -rf $O/stats16 $O/pmc_fetch16 $O/pmc_write16
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
cp $O/fn2stats/*/*kernel_stats.csv $O/flownet2_kernel_stats.csv
rm -rf $O/stats $O/fn2stats $O/pmc_fetch $O/pmc_write $O/pmc_mfma
tail -n 1 $O/bench.json
