#!/bin/bash
run() { python bench.py --no-variants --no-cpu-baseline --steps 50 --warmup 10 $2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'])"; }
export MGNNS_LSTM_GRID=128 MGNNS_SCHEDULE=channels MGNNS_GRAPH_MODE=segments
for ro in 0 1; do for nq in 0 1; do
  export MGNNS_TAIL_READOUT=$ro MGNNS_TAIL_NEXTQ=$nq
  run "readout_in=$ro nextq_in=$nq B=256"
  run "readout_in=$ro nextq_in=$nq B=32" "--batch 32"
done; done
export MGNNS_FUSED_LABEL_TAIL=0
run "unfused B=256"; run "unfused B=32" "--batch 32"
