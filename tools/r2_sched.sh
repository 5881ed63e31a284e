#!/bin/bash
# scratch experiment script (GPU box)
mkdir -p gpurun_out
one() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'], l.get('max_abs_logit_diff_vs_cpu_oracle'), l['config'].get('collective'))"; }
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -k "comm or world1 or init_all or classifier or forward or label_tail" 2>&1 | tail -6
for i in 1 2; do
timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one head-fused
MGNNS_FUSED_HEAD=0 timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one head-3launch
done
for b in 96 128; do
timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one B$b-cluster
MGNNS_LABEL_TAIL_CLUSTER=0 timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline --steps 30 2>/dev/null | one B$b-single
done
MGNNS_FORCE_DIST=1 MGNNS_COLLECTIVE=abi timeout 600 python bench.py --no-variants --no-cpu-baseline > gpurun_out/abi.json 2> gpurun_out/abi.err; echo "abi rc=$?"; tail -c 600 gpurun_out/abi.err; cat gpurun_out/abi.json | one abi-dist1
MGNNS_FORCE_DIST=1 timeout 600 python bench.py --no-variants --no-cpu-baseline 2>/dev/null | one torch-dist1
