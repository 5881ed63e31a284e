#!/bin/bash
# scratch experiment script (GPU box)
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['config']['launch'])"; }
timeout 300 python tools/dev/bench_label_gcn.py 2>&1 | tail -24
for gr in 32 64 128; do
MGNNS_LGCN_GRID=$gr timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one grid$gr
MGNNS_LGCN_GRID=$gr timeout 300 python bench.py --batch 32 --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one B32-grid$gr
done
