#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export MGNNS_GRAPH_MODE=segments
rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o pk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants > /tmp/pk.log 2>&1
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/pk/pk_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'classifier_head' in r['Kernel_Name']]
h = idx[-8]; prev = idx[-9]
t0 = int(rows[prev]['End_Timestamp'])
for r in rows[prev + 1:h + 1]:
    print("%8.1f %7.1f q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], r['Kernel_Name'][24:70]))
PY
