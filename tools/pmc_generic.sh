#!/bin/bash
# usage: bash tools/pmc_generic.sh <outdir-name> <python script + args ...>   -- two SQ counter passes + per-kernel summary
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -- python $R/"$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -- python $R/"$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/summary.json $O/p1/*/*counter_collection.csv $O/p2/*/*counter_collection.csv
rm -rf $O/p1 $O/p2
