#!/usr/bin/env python
"""ms per forward (one hipGraph replay at a time) of small shards of the bench workload under different schedules:
    python tools/dev/small_batch_time.py [schedule ...]      (default: auto small small2)"""
import os
import statistics
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
model.set_precision("bf16").set_attention("faithful")
scheds = sys.argv[1:] or ["auto", "small", "small2"]
ref = {}
with torch.no_grad():
    for bs in (32, 64, 128):
        sub = {k: (v[:bs] if k != "label_query" else v) for k, v in inp.items()}
        call = harness.call_args(sub, dev)
        for name in scheds:
            model.schedule = name
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
            out = gf.static_out[:bs].float().cpu()
            ref.setdefault(bs, out)
            print("B=%3d %-10s %.4f ms (min %.4f)  max |dlogit| vs first schedule %.2e" % (bs, name, statistics.median(ds), min(ds), float((out - ref[bs]).abs().max())), flush=True)
            del gf
