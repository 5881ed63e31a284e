#!/bin/bash
run() { python bench.py --no-variants --no-cpu-baseline --steps 50 --warmup 10 $2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['max_abs_logit_diff_vs_cpu_oracle'])"; }
export MGNNS_GRAPH_MODE=segments
MGNNS_SCHEDULE=channels run "channels"
for sh in 4 3 2; do for g in 64 128; do
  export MGNNS_SCHEDULE=masked MGNNS_LSTM_CU_SHARE=$sh MGNNS_LSTM_GRID=$g
  run "masked share=1/$sh lstm_grid=$g"
done; done
export MGNNS_SCHEDULE=masked MGNNS_LSTM_CU_SHARE=4 MGNNS_LSTM_GRID=64
python tools/graph_timeline.py 2>&1 | tail -24
python -m pytest tests/test_model_gpu.py -q -x -k "graph_replay or golden_logits" 2>&1 | tail -2
