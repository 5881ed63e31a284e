for r in 1 2; do
for v in default g160nomma g160stream; do
  if [ $v = default ]; then unset MGNNS_LIB; else export MGNNS_LIB=mgnns_amd/variants/lib_$v.so; fi
  echo "-- round $r $v"; python tools/dev/gemm_time.py 1024 2>/dev/null | grep "160 x 256:" | head -2
done; done
