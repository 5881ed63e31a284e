#!/bin/bash
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "fused_layer" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -3
run() { timeout 300 python bench.py --no-variants --steps 50 --warmup 10 $2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['max_abs_logit_diff_vs_cpu_oracle'])"; }
MGNNS_FUSED_LAYER=1 run "fused layer"
MGNNS_FUSED_LAYER=0 run "separate core + tail"
MGNNS_FUSED_LAYER=1 timeout 200 python tools/graph_timeline.py 2>&1 | tail -22
