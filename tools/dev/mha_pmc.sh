#!/bin/bash
# usage (GPU box, repo root): [MGNNS_LIB=...] tools/dev/mha_pmc.sh <tag>   -- SQ counter passes over sq_mha_core_bf16 alone
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $root/gpurun_out/mhapmc_${tag}_$i -o p -- python3 $root/tools/dev/mha_prof.py > $root/gpurun_out/mhapmc_${tag}_$i.log 2>&1
  db=$(find $root/gpurun_out/mhapmc_${tag}_$i -name "*.db" | head -1)
  python3 $root/tools/rocpd_pmc.py $db sq_mha_core
  rm -rf $root/gpurun_out/mhapmc_${tag}_$i
done
