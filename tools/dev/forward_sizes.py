#!/usr/bin/env python
"""hipGraph replay time of the whole forward at B = 256 / 128 / 64 / 32 (bf16 mode): python tools/dev/forward_sizes.py [reps]
A/B builds of one kernel: MGNNS_LIB=mgnns_amd/variants/lib_x.so (tools/dev/build_variant.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import harness, synth  # noqa: E402
from mgnns_amd.graph import GraphedForward  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.CONFIGS["mvsa_multiple_b256"]
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=256, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision(os.environ.get("PRECISION", "bf16"))
if os.environ.get("SCHEDULE"):
    model.schedule = os.environ["SCHEDULE"]
sizes = [int(x) for x in os.environ.get("SIZES", "256,128,64,32").split(",")]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.no_grad():
    for bs in sizes:
        sub = {k: (v[:bs] if k != "label_query" else v) for k, v in inp.items()}
        gv = GraphedForward(model, harness.call_args(sub, dev))
        best = []
        for _ in range(reps):
            for _ in range(10):
                gv.replay()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(40):
                gv.replay()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t) / 40 * 1e3)
        print("B=%3d: %s ms" % (bs, " ".join("%.4f" % x for x in best)), flush=True)
