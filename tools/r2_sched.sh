#!/bin/bash
run() { timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 50 --warmup 10 $2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'])"; }
export MGNNS_GRAPH_MODE=segments
for sch in channels banks_serial; do MGNNS_SCHEDULE=$sch run "$sch"; done
MGNNS_SCHEDULE=banks_serial timeout 200 python tools/graph_timeline.py 2>&1 | tail -22
