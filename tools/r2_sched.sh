#!/bin/bash
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "mha or fused_layer" > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -5
for v in v2 v1 v2 v1; do
if [ $v = v1 ]; then export MGNNS_LIB=mgnns_amd/variants/lib_v1.so; else unset MGNNS_LIB; fi
echo "== $v"; timeout 200 python tools/bench_kernels.py mha_bf16 2>&1 | grep "sq_mha" | head -3
done
