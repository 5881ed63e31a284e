#!/bin/bash
# usage (GPU box, repo root): tools/dev/spmm_pmc.sh <tag> <graph> <F> <path> <variant>  -- separate --pmc passes, kernel-trace only
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --kernel-trace -d $root/gpurun_out/pmc_${tag}_$name -o p -- python3 $root/tools/dev/spmm_prof.py "$@" > $root/gpurun_out/pmc_${tag}_$name.log 2>&1
  db=$(ls $root/gpurun_out/pmc_${tag}_$name/*/*.db 2>/dev/null | head -1)
  [ -z "$db" ] && db=$(ls $root/gpurun_out/pmc_${tag}_$name/*.db 2>/dev/null | head -1)
  echo "== $tag $ctr ($db)"
  python3 $root/tools/rocpd_pmc.py $db spmm
done
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/kt_${tag} -o p -- python3 $root/tools/dev/spmm_prof.py "$@" > $root/gpurun_out/kt_${tag}.log 2>&1
f=$(find $root/gpurun_out/kt_${tag} -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -5 $f
