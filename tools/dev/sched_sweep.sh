mkdir -p gpurun_out/f8
for s in channels banks_first banks_serial channels2 bigsmall tails_first; do
  MGNNS_SCHEDULE=$s python bench.py --no-variants --no-cpu-baseline --steps 30 > gpurun_out/f8/$s.txt 2>&1
done
