#!/bin/bash
# scratch experiment script (GPU box)
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['config']['launch'])"; }
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "classifier" 2>&1 | tail -3
for i in 1 2 3; do
timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one default
done
timeout 300 python bench.py --batch 32 --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one B32
timeout 300 python bench.py --batch 64 --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one B64
timeout 300 python bench.py --batch 128 --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one B128
