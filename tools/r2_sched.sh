#!/bin/bash
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['config']['launch'][:40])"; }
for i in 1 2; do
for b in 256 128; do
for s in channels channels2; do
MGNNS_SCHEDULE=$s MGNNS_GRAPH_MODE=segments timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline --steps 50 --warmup 20 2>gpurun_out/s.err | one B$b-$s-seg
done
done
done
timeout 300 python bench.py --no-variants --no-cpu-baseline 2>gpurun_out/s.err | one default
timeout 300 python bench.py --batch 32 --no-variants --no-cpu-baseline 2>gpurun_out/s.err | one default-B32
