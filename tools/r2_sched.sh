#!/bin/bash
timeout 600 python -m pytest tests/test_stress_gpu.py tests/test_ops_gpu.py -x -q -k "stress or spmm" > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -3
timeout 300 python tools/bench_stress.py 2>/dev/null | tail -1 > gpurun_out/r02_stress_gcn.json; python -c "
import json; d=json.load(open('gpurun_out/r02_stress_gcn.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'cold_ms' in v: print(k, v['cold_ms'], v.get('cold_GBps'), v.get('frac_of_8TBps'), v.get('warm_ms_same_buffers'), v.get('copy_cold_GBps'))
"
