#!/bin/bash
# Experimental build of ONE source with extra defines, linked against the product objects:
#   bash tools/build_variant.sh vv_wgrad_bf16 "-DVV_EXPR=1" exp_libs/lib_r1.so
# (run the product with VV_LIB_PATH=<that .so>; exp_libs/ is git-ignored and travels with gpurun)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; DEF=$2; OUT=$3
mkdir -p $R/exp_libs /tmp/vvvar
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $DEF -I$R/include -c $R/vec_vad_amd/csrc/$SRC.hip -o /tmp/vvvar/$SRC.$$.o 2> /dev/null
OBJS=$(ls $R/vec_vad_amd/csrc/build/*.hip.o | grep -v "/$SRC.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/$OUT $OBJS /tmp/vvvar/$SRC.$$.o
rm -f /tmp/vvvar/$SRC.$$.o
echo built $OUT
