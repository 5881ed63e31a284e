#!/bin/bash
# usage (GPU box): tools/dev/imgbank_pmc.sh <lib or ""> : FETCH_SIZE / WRITE_SIZE of imgbank_pool_bf16 + its time
root=${GRAFT_REPO_ROOT:-$(pwd)}
lib=$1
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export MGNNS_LIB=$root/$lib
python3 $root/tools/bench_kernels.py imgbank 2>/dev/null | grep bf16
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ipmc; rocprofv3 --pmc $ctr --kernel-trace -d /tmp/ipmc -o p -- python3 $root/tools/bench_kernels.py imgbank > /dev/null 2>&1
  python3 $root/tools/rocpd_pmc.py $(find /tmp/ipmc -name "*.db" | head -1) imgbank_pool_bf16
done
