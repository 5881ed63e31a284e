#!/bin/bash
# r6: the 160 x 256 / 320 x 256 tiles with 64-wide K slices (default) against 32-wide (MGNNS_GEMM160_BK=32 / MGNNS_GEMM320_BK=32), cache-cold,
# alternating; tests first
out=gpurun_out/r6_gemm_bk2.txt
: > $out
python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_bf16" 2>&1 | tail -2 >> $out
python -m pytest tests/test_stress_gpu.py -x -q -k "full_size and dense" 2>&1 | tail -2 >> $out
for r in 1 2 3; do
  for bk in 32 64; do
    echo "-- round $r MGNNS_GEMM320_BK=$bk" >> $out
    MGNNS_GEMM320_BK=$bk python tools/dev/gemm_time.py 2048 2>/dev/null | grep "320 x 256:\|by estimate" | head -3 >> $out
  done
done
for bk in 32 64; do echo "-- dense workload launches, MGNNS_GEMM320_BK=$bk" >> $out; MGNNS_GEMM320_BK=$bk python tools/dev/stress_launches.py 3 2>/dev/null | grep "default pick" >> $out; done
cat $out
