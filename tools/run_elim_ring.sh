for m in 0 1; do
for v in base e1 e2 e4 e8 e16 e32 e64 e3 e65 e12; do
  if [ $v = base ]; then lib=""; else lib="exp_libs/ring_$v.so"; fi
  echo "== plain=$m variant $v"
  UB_PLAIN=$m UB_ONLY32=1 VV_LIB_PATH=$lib timeout 120 python tools/ubench_wino.py 20 2>&1 | grep "H=32" | sed -e 's/exec.*direct/|/' -e 's/TF\/s | err.*per-tile/| per-tile/'
done; done
