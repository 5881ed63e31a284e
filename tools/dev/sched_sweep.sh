#!/bin/bash
# usage (GPU box, repo root): tools/dev/sched_sweep.sh [bench args]   -- the default bench line under every schedule -> gpurun_out/sched/
mkdir -p gpurun_out/sched
for s in ${SCHEDS:-channels banks_first banks_serial channels2 bigsmall tails_first place_bank_first place_bank_first_lgcn_s3}; do
  MGNNS_SCHEDULE=$s python bench.py --no-variants --no-cpu-baseline --steps 30 "$@" > gpurun_out/sched/$s.txt 2>&1
done
