#!/bin/bash
export MGNNS_LIB=mgnns_amd/variants/lib_spmmexp.so
for e in "16 2 1 384" "8 4 1 384" "8 4 1 256" "8 2 1 384" "4 8 1 256" "4 4 1 384" "8 8 1 256" "8 4 2 256" "16 4 1 384"; do
MGNNS_SPMM_EXP="$e" timeout 100 python tools/dev/spmm_exp.py 2>&1 | tail -1
done
