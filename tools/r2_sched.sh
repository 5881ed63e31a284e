#!/bin/bash
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['config']['launch'][:40])"; }
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "schedule" 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
for b in 256 128; do
for s in channels channels_m; do
MGNNS_SCHEDULE=$s timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline --steps 50 --warmup 10 2>gpurun_out/s.err | one B$b-$s || tail -3 gpurun_out/s.err
done
done
done
