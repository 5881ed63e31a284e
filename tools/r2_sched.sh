#!/bin/bash
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -s -k "label_tail" 2>&1 | grep -E "label_tail_bf16|passed|failed|Error" | tail -14
timeout 600 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -3
run() { timeout 400 python bench.py --no-variants --steps 50 --warmup 10 $2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['max_abs_logit_diff_vs_cpu_oracle'])"; }
export MGNNS_GRAPH_MODE=segments
MGNNS_LABEL_TAIL_TERMS=3 run "bf16 fused tail terms=3"
MGNNS_LABEL_TAIL_TERMS=1 run "bf16 fused tail terms=1"
MGNNS_FUSED_LABEL_TAIL_BF16=0 run "fp32 fused tail"
