#!/bin/bash
# item 4, second sweep: cluster sizes of the tails (fewer, longer workgroups) and three forwards in flight
out=gpurun_out/r6_thrifty2.txt
: > $out
run() {
  name=$1; extra=$2; shift 2
  line=$(env "$@" python bench.py --no-variants --no-cpu-baseline --steps 30 --warmup 5 $extra 2>/dev/null | tail -1)
  python - "$name" "$line" >> $out <<'P'
import json, sys
l = json.loads(sys.argv[2])
print("%-28s %.4f ms in flight  %.4f ms one at a time" % (sys.argv[1], l["ms_per_step"], l.get("ms_per_step_one_in_flight", float("nan"))))
P
}
for r in 1 2 3; do
  echo "-- round $r" >> $out
  run default "" X=1
  run label_tail_cluster=off "" MGNNS_LABEL_TAIL_CLUSTER=0
  run mha_tail_cluster=1 "" MGNNS_TAIL_CLUSTER=1
  run mha_tail_cluster=2 "" MGNNS_TAIL_CLUSTER=2
  run three_in_flight "--in-flight 3" X=1
done
cat $out
