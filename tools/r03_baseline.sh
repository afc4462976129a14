#!/bin/bash
# Round-3 starting measurements (run on the GPU box):
#   gpurun --timeout 1500 -- 'bash tools/r03_baseline.sh'
# PMC traffic + MFMA-busy for BASELINE config 4 (Full, B=512, mixed bf16) and config 5 (FlowNet2), the per-launch breakdown of
# config 4, and the small-batch regime (B=32 / B=16 per rank) of the eager launch loop.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_base
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
C4="$B --precision bf16 --model full --batch 512"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $C4 --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $C4 --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_mfma -- $C4 --steps 3 --warmup 2 > /dev/null 2>&1
python $R/tools/make_pmc_mfma.py $O/pmc_mfma/*/*counter_collection.csv $O/pmc_mfma_busy_bf16_full_b512.json > $O/mfma16.txt
python $R/tools/make_pmc_traffic.py $O/pmc_fetch/*/*counter_collection.csv $O/pmc_write/*/*counter_collection.csv $O/pmc_hbm_traffic_bf16_full_b512.json > $O/traffic16.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
F="python $R/tools/bench_flownet2.py --eager"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $F > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $F > /dev/null 2>&1
python $R/tools/make_pmc_traffic.py $O/pmc_fetch/*/*counter_collection.csv $O/pmc_write/*/*counter_collection.csv $O/pmc_hbm_traffic_flownet2.json > $O/traffic_fn2.txt
rm -rf $O/pmc_fetch $O/pmc_write
$C4 --steps 10 --warmup 3 --breakdown > $O/bench_c4.json 2> $O/bd_c4.txt
for b in 16 32 64; do
  $B --batch $b --steps 50 --warmup 10 > $O/bench_b$b.json 2> $O/bench_b$b.err
done
rocprofv3 --kernel-trace --output-format csv -d $O/kt32 -- $B --batch 32 --steps 12 --warmup 3 > /dev/null 2>&1
python $R/tools/step_gaps.py $O/kt32/*/*kernel_trace.csv > $O/gaps_b32.txt
rm -rf $O/kt32
python - <<EOF
import json
for b in (16, 32, 64):
    try:
        d = json.load(open('$O/bench_b%d.json' % b))
        print('B=%d  %.0f cubes/s  %.3f ms/step' % (b, d['value'], d['ms_per_step']))
    except Exception as e:
        print('B=%d failed' % b, e)
d = json.load(open('$O/bench_c4.json'))
print('c4', d['value'], d['ms_per_step'], d['roofline']['frac'])
EOF
tail -5 $O/gaps_b32.txt
cat $O/traffic16.txt
