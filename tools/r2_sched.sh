#!/bin/bash
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['config']['launch'][:50])"; }
timeout 900 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -3
for b in 256 32; do
timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline 2>gpurun_out/s.err | one B$b
done
