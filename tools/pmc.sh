#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <tag> <counter> <bench args...>   -- ONE counter set per pass, kernel-trace only
tag=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace -d $root/gpurun_out/pmc_$tag -o $tag -- python3 $root/bench.py "$@" > $root/gpurun_out/pmc_$tag.log 2>&1
grep -c . $root/gpurun_out/pmc_$tag.log
ls $root/gpurun_out/pmc_$tag
