# usage: prof_cmp.sh <libname or ""> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$1" ] && export VV_LIB_PATH=$R/exp_libs/$1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$2 -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $R/gpurun_out/st_$2.json 2>/dev/null
cp $R/gpurun_out/st_$2/*/*kernel_stats.csv $R/gpurun_out/st_$2.csv; rm -rf $R/gpurun_out/st_$2
