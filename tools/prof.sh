#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> <bench args...>
# rocprofv3 kernel trace of bench.py -> gpurun_out/prof_<tag>/ (rocpd db) + gpurun_out/prof_<tag>.log
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
grep metric $root/gpurun_out/prof_$tag.log | cut -c1-220
