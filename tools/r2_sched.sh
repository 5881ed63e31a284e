#!/bin/bash
mkdir -p gpurun_out
MGNNS_FORCE_DIST=1 timeout 600 python bench.py --no-variants --no-cpu-baseline > gpurun_out/d1.json 2> gpurun_out/d1.err; echo "dist1 rc=$?"; tail -c 300 gpurun_out/d1.err; cut -c1-700 gpurun_out/d1.json
MGNNS_BENCH_BACKEND=gloo MGNNS_BENCH_SAME_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/two.json 2> gpurun_out/two.err; echo "two-rank rc=$?"; tail -c 300 gpurun_out/two.err; python -c "
import json; l=json.loads(open('gpurun_out/two.json').read().strip().splitlines()[-1]); print(l['n_gpus'], l['value'], l['ms_per_step'], l['config']['launch'], l.get('weak_scaling',{}).get('ms_per_step'), l.get('strong_scaling',{}).get('ms_per_step'), l.get('strong_scaling',{}).get('per_gpu_batch'))"
