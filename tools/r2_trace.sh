#!/bin/bash
# usage: tools/r2_trace.sh <tag> [env assignments...]  -- short kernel trace of bench.py graph replays -> gpurun_out/prof_<tag>/
tag=$1; shift
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --steps 8 --warmup 3 --no-variants --no-cpu-baseline $BENCH_ARGS > $root/gpurun_out/prof_$tag.log 2>&1
grep -o '"value": [0-9.]*' $root/gpurun_out/prof_$tag.log | head -1
cd $root && python3 tools/trace_timeline.py $(ls gpurun_out/prof_$tag/*/*.db gpurun_out/prof_$tag/*.db 2>/dev/null | head -1) > gpurun_out/timeline_$tag.txt 2>&1
head -3 gpurun_out/timeline_$tag.txt
