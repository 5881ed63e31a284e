#!/bin/bash
# scratch script for one-off GPU-box experiments (gpurun -- 'timeout 800 tools/r2_sched.sh'); see tools/r2_check.sh and
# tools/round_profiles.sh for the reproducible runs
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l['config']['launch'][:50])"; }
for b in 256 128 64 32; do
timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline 2>gpurun_out/s.err | one B$b
done
