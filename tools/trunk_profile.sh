#!/bin/bash
# usage (GPU box, repo root): tools/trunk_profile.sh <tag>  -- rocprofv3 kernel stats of the ResNet-101 trunk bench
# -> gpurun_out/prof_<tag>_trunk/ (rocpd db), gpurun_out/<tag>_trunk_kernel_stats.md, gpurun_out/<tag>_trunk_bench.txt
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $root/tools/bench_trunk.py --batch 128 --per-layer > $root/gpurun_out/${tag}_trunk_bench.txt 2>&1
python3 $root/tools/bench_trunk.py --batch 128 --arch resnet50 >> $root/gpurun_out/${tag}_trunk_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_${tag}_trunk -o ${tag}_trunk -- python3 $root/tools/bench_trunk.py --batch 128 --iters 3 > $root/gpurun_out/prof_${tag}_trunk.log 2>&1
cd $root
python3 - <<PY
import glob, sys
sys.path.insert(0, "tools")
import write_profiles as W
db = sorted(glob.glob("gpurun_out/prof_${tag}_trunk/**/*.db", recursive=True))[-1]
open("gpurun_out/${tag}_trunk_kernel_stats.md", "w").write(
    "# rocprofv3 --kernel-trace --stats: python3 tools/bench_trunk.py --batch 128 --iters 3 (ResNet-101 features, 448x448, 5 passes incl. warm-up)\n\n"
    + W.kernel_stats(db)[0] + "\n")
PY
tail -3 $root/gpurun_out/${tag}_trunk_bench.txt
