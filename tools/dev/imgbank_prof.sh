#!/bin/bash
# usage (GPU box, repo root): tools/dev/imgbank_prof.sh <tag> [lib.so]   -> kernel durations of imgbank_pool_bf16 under rocprofv3
tag=$1; lib=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export MGNNS_LIB=$root/$lib
rocprofv3 --kernel-trace --stats -d /tmp/ip_$tag -o $tag -- python3 $root/tools/dev/imgbank_time.py > /tmp/ip_$tag.log 2>&1
python3 $root/tools/rocpd_stats.py $(find /tmp/ip_$tag -name "*.db" | head -1) 2>/dev/null | grep -i "imgbank" | head -3
python3 $root/tools/dev/rocpd_seq.py $(find /tmp/ip_$tag -name "*.db" | head -1) imgbank_pool_bf16 36
tail -1 /tmp/ip_$tag.log
