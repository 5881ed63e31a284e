#!/bin/bash
# ms per forward (two in flight / one at a time) of schedules x batches [x GPU_MAX_HW_QUEUES]: tools/dev/queues_sweep.sh "32 64" "auto small_a small_b" ["0 8"]
for b in ${1:-32 64 256}; do for q in ${3:-0}; do for s in ${2:-auto}; do
  if [ $q = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  MGNNS_SCHEDULE=$s timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$b queues=$q schedule=$s', d['ms_per_step'], d.get('ms_per_step_one_in_flight'))
"; done; done; done
