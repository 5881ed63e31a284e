#!/usr/bin/env python
"""ms per forward (one hipGraph replay at a time) of small shards of the bench workload under launch-shape knobs set in process:
    python tools/dev/small_knobs.py lgcn_grid=0,128,192,256 [batches=32,64]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import harness, ops, synth  # noqa: E402
from mgnns_amd.graph import GraphedForward  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.CONFIGS["mvsa_multiple_b256"]
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=256, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision("bf16").set_attention("faithful")
knobs = dict(a.split("=") for a in sys.argv[1:])
batches = [int(x) for x in knobs.pop("batches", "32,64").split(",")]
(name, vals), = knobs.items() if knobs else (("lgcn_grid", "0"),)
vals = [int(v) for v in vals.split(",")]


def setk(v):
    if name == "lgcn_grid":
        ops.LABEL_GCN_GRID = v
    elif name == "tail_cluster":
        ops.LABEL_TAIL_CLUSTER = v
    else:
        raise SystemExit("unknown knob " + name)


with torch.no_grad():
    for bs in batches:
        sub = {k: (v[:bs] if k != "label_query" else v) for k, v in inp.items()}
        call = harness.call_args(sub, dev)
        for rep in range(2):
            for v in vals:
                setk(v)
                gf = GraphedForward(model, call)
                for _ in range(10):
                    gf.replay()
                torch.cuda.synchronize()
                ds = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(30):
                        gf.replay()
                    torch.cuda.synchronize()
                    ds.append((time.perf_counter() - t0) / 30 * 1e3)
                print("B=%d %s=%d: %.4f ms (min %.4f)" % (bs, name, v, statistics.median(ds), min(ds)), flush=True)
                del gf
