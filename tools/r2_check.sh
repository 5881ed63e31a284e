#!/bin/bash
# usage (GPU box, repo root): tools/r2_check.sh <tag>  -- GPU tests + default bench + 1-GPU RCCL path + batch sweep
tag=${1:-r02a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_pytest.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/${tag}_bench.err; cut -c1-1500 gpurun_out/${tag}_bench.json
MGNNS_FORCE_DIST=1 timeout 600 python bench.py --no-variants --no-cpu-baseline > gpurun_out/${tag}_bench_dist1.json 2> gpurun_out/${tag}_bench_dist1.err; echo "dist1 rc=$?"; tail -c 800 gpurun_out/${tag}_bench_dist1.err; cut -c1-900 gpurun_out/${tag}_bench_dist1.json
# the whole N-rank code path on this one GPU (gloo, both ranks on cuda:0; the SECOND scaling measured is slow there -- two
# processes time-slicing one GPU with a host-side gather per step -- so only the protocol / n_gpus / shapes are checked)
MGNNS_BENCH_BACKEND=gloo MGNNS_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/${tag}_two_ranks.json 2> gpurun_out/${tag}_two_ranks.err; echo "two-rank rc=$?"; cut -c1-300 gpurun_out/${tag}_two_ranks.json
for b in 32 64 128; do
  timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b', l['value'], l['ms_per_step'])"
done
