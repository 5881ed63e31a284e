#!/bin/bash
run() { timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 50 --warmup 10 $2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['config']['launch'])"; }
for mb in 96 1; do for b in 32 64; do MGNNS_FUSED_TAIL_MIN_BATCH=$mb run "min_batch=$mb B=$b" "--batch $b"; done; done
