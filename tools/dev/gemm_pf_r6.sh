#!/bin/bash
# r6: the 160 x 256 dense GEMM with the compute waves touching the operand lines MG_GEMM160_PF slices ahead (an L2 prefetch), cache-cold,
# three alternating rounds on one box; correctness of the variants first
out=gpurun_out/r6_gemm_pf.txt
: > $out
for v in default g160pf4 g160pf8 g160pf16; do
  if [ $v = default ]; then unset MGNNS_LIB; else export MGNNS_LIB=mgnns_amd/variants/lib_$v.so; fi
  echo "-- tests $v" >> $out; python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_bf16" 2>&1 | tail -1 >> $out
done
for r in 1 2 3; do
  for v in default g160pf4 g160pf8 g160pf16; do
    if [ $v = default ]; then unset MGNNS_LIB; else export MGNNS_LIB=mgnns_amd/variants/lib_$v.so; fi
    echo "-- round $r $v" >> $out; python tools/dev/gemm_time.py 1024 2>/dev/null | grep -i "forced\|160\|F=" | head -4 >> $out
  done
done
cat $out
