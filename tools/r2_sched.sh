#!/bin/bash
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l.get('max_abs_logit_diff_vs_cpu_oracle'))"; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "lstm or forward or graph or pipeline" > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -4
for i in 1 2; do
for b in 32 64 256; do
timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline --steps 50 --warmup 10 2>gpurun_out/s.err | one B$b
done
done
timeout 200 python tools/bench_kernels.py lstm 2>&1 | tail -6
