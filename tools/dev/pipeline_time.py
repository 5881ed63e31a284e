#!/usr/bin/env python
"""Throughput of back-to-back forwards: one GraphedForward (join between forwards) vs GraphedPipeline (two in flight)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import harness, synth  # noqa: E402
from mgnns_amd.graph import GraphedForward, GraphedPipeline  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.CONFIGS["mvsa_multiple_b256"]
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=256, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision("bf16")
sizes = [int(x) for x in os.environ.get("SIZES", "256,128,32").split(",")]
with torch.no_grad():
    for bs in sizes:
        sub = {k: (v[:bs] if k != "label_query" else v) for k, v in inp.items()}
        args = harness.call_args(sub, dev)
        gv = GraphedForward(model, args)
        ref = gv.replay().clone()
        torch.cuda.synchronize()
        res = {}
        for depth in (1, 2, 3):
            pipe = GraphedPipeline(model, args, depth=depth)
            for _ in range(12):
                it = pipe.replay()
            pipe.wait()
            torch.cuda.synchronize()
            for it in pipe.items:
                assert torch.equal(it.static_out, ref), "pipelined logits differ"
            t = time.perf_counter()
            for _ in range(60):
                pipe.replay()
            torch.cuda.synchronize()
            res[depth] = (time.perf_counter() - t) / 60 * 1e3
            del pipe
        t = time.perf_counter()
        for _ in range(60):
            gv.replay()
        torch.cuda.synchronize()
        base = (time.perf_counter() - t) / 60 * 1e3
        print("B=%3d: serial replay %.4f ms | pipeline depth 1 / 2 / 3: %.4f / %.4f / %.4f ms per forward" % (bs, base, res[1], res[2], res[3]), flush=True)
