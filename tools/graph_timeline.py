#!/usr/bin/env python
"""REAL timeline of one hipGraph replay of the B=256 forward: one-thread stamp kernels (mgnns_debug_stamp, the GPU's
100 MHz real-time counter) are captured into the graph at the start / end of every channel and fusion stack, on the
stream that runs it.  rocprofv3 perturbs the concurrency of the four branches; this does not (each stamp is one extra
~2 us launch on its stream).

    python tools/graph_timeline.py [--dtype bf16|f32] [--attn faithful|folded]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import harness, ops, synth  # noqa: E402
from mgnns_amd.graph import GraphedForward  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--attn", default="faithful")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--schedule", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synth.CONFIGS["mvsa_multiple_b256"]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp = synth.make_inputs(cfg, B=args.batch, seed=cfg.seed, pmi=pmi)
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
    model.set_precision("bf16" if args.dtype == "bf16" else "fp32").set_attention(args.attn)
    if args.schedule:
        model.schedule = args.schedule
    call = harness.call_args(inp, dev)
    with torch.no_grad():
        model(*call)                                  # weight packing etc. outside the recording
        torch.cuda.synchronize()
        slots, names = ops.timeline_begin(dev)
        gf = GraphedForward(model, call, warmup=0)    # capture (stamps included)
        ops.timeline_end()
        for _ in range(5):
            gf.replay()
        torch.cuda.synchronize()
    t = slots.cpu().tolist()
    t0 = min(t[i] for i in range(len(names)))
    order = sorted(range(len(names)), key=lambda i: t[i])
    print("us     event")
    for i in order:
        print("%6.1f  %s" % ((t[i] - t0) / 100.0, names[i]))


if __name__ == "__main__":
    main()
