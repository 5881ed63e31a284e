#!/bin/bash
# do FEWER CUs for the latency-bound persistent kernels raise the two-in-flight throughput?
# ms two in flight / one at a time of the headline forward per setting, three alternating rounds on one box.
out=gpurun_out/r6_thrifty.txt
: > $out
run() {  # name, env...
  name=$1; shift
  line=$(env "$@" python bench.py --no-variants --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1)
  python - "$name" "$line" >> $out <<'P'
import json, sys
l = json.loads(sys.argv[2])
print("%-28s %.4f ms two in flight  %.4f ms one at a time" % (sys.argv[1], l["ms_per_step"], l.get("ms_per_step_one_in_flight", float("nan"))))
P
}
for r in 1 2 3; do
  echo "-- round $r" >> $out
  run default X=1
  run lgcn_grid=32 MGNNS_LGCN_GRID=32
  run lgcn_grid=48 MGNNS_LGCN_GRID=48
  run lstm_grid=64 MGNNS_LSTM_GRID=64
  run lstm_grid=96 MGNNS_LSTM_GRID=96
  run lgcn32+lstm96 MGNNS_LGCN_GRID=32 MGNNS_LSTM_GRID=96
done
cat $out
