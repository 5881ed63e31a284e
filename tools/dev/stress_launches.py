#!/usr/bin/env python
"""configs[4], dense adjacency: every launch of the eager 3-channel forward between HIP events, round 4's GEMM forms
(gemm_bf16_set_form(0)) against the default pick, and the (adj . X) . W variant.
    python tools/dev/stress_launches.py [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops, stress  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
res = {}
for order in ("reference", "reassociated"):
    wl = stress.StressWorkload(dense=True, order=order)
    for form, tag in ((0, "round-4 forms"), (-1, "default pick")):
        ops.gemm_bf16_set_form(form)
        for _ in range(2):
            wl.forward(union=False)
        rows = None
        for _ in range(reps):
            with stress.launch_times() as lt:
                wl.forward(union=False)
            ms = [v for _, v in lt.ms]
            rows = ms if rows is None else [min(a, b) for a, b in zip(rows, ms)]
        labels = [k for k, _ in lt.ms]
        whole = stress.time_warm(lambda: wl.forward(union=False), (), reps=5)
        res["%s / %s" % (order, tag)] = {"launches_ms_min_of_%d" % reps: list(zip(labels, rows)), "sum_ms": round(sum(rows), 4),
                                         "forward_ms_eager_back_to_back": round(whole, 4)}
        print("%-12s %-14s sum %.4f ms, forward %.4f ms: %s" % (order, tag, sum(rows), whole,
              "  ".join("%s %.3f" % (k, v) for k, v in zip(labels[:5], rows[:5]))), flush=True)
    ops.gemm_bf16_set_form(-1)
    del wl
    torch.cuda.empty_cache()
print(json.dumps(res))
