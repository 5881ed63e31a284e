for b in 64 96 128 192; do for l in 1 0 1 0; do
  MGNNS_TEXTGCN_LEAN=$l timeout 300 python bench.py --batch $b --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$b lean=$l', d['ms_per_step'], d.get('ms_per_step_one_in_flight'))
"; done; done
