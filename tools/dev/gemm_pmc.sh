#!/bin/bash
# usage (GPU box, repo root): tools/dev/gemm_pmc.sh <tag> <F> [launches]  -- separate --pmc passes of the dense bf16 GEMM, kernel-trace only
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum"; do
  name=$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${tag}_$name -o p -- python3 $root/tools/dev/gemm_prof.py "$@" > /tmp/pmc_${tag}_$name.log 2>&1
  db=$(find /tmp/pmc_${tag}_$name -name "*.db" | head -1)
  echo "== $tag $ctr"
  python3 $root/tools/rocpd_pmc.py $db gemm_bf16_nt
done
