#!/bin/bash
# elimination builds of wino44_conv_kernel (VV_EXP4 bit mask; every value other than 0 computes wrong results): where the time goes
for v in base e1 e2 e4 e8 e6 e7; do
  if [ $v = base ]; then lib=""; else lib="exp_libs/w44_$v.so"; fi
  echo "== variant $v"
  VV_LIB_PATH=$lib timeout 120 python tools/ubench_wino.py 10 2>&1 | grep -o "^.\{20\} H=.\{12\}\|F(4x4) *[0-9.]* us" | paste - - 
done
