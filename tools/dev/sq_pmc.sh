#!/bin/bash
# usage (GPU box, repo root): tools/dev/sq_pmc.sh <tag> <kernel-substring> <graph> <F> <path> <variant>   -- one SQ counter pass
tag=$1; sub=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $root/gpurun_out/sq_$tag -o p -- python3 $root/tools/dev/spmm_prof.py "$@" > $root/gpurun_out/sq_$tag.log 2>&1
db=$(find $root/gpurun_out/sq_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_pmc.py $db $sub
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM --kernel-trace -d $root/gpurun_out/sq2_$tag -o p -- python3 $root/tools/dev/spmm_prof.py "$@" > $root/gpurun_out/sq2_$tag.log 2>&1
db=$(find $root/gpurun_out/sq2_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_pmc.py $db $sub
