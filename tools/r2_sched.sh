#!/bin/bash
# scratch experiment script (GPU box)
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l.get('max_abs_logit_diff_vs_cpu_oracle'))"; }
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -k "schedule or persistent or graph_replay" 2>&1 | tail -6
for b in 16 32 64 128 256; do
timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline --steps 30 2>gpurun_out/s.err | one B$b
done
