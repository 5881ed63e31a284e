#!/usr/bin/env python
"""Upper bound of anything the TEXT side (BiLSTM chains, text GCN, masked attention launches) can return to the headline: the same
B = 256 forward with every document cut to `n` tokens (the image side untouched), two in flight / one at a time, alternating rounds.
    python tools/dev/text_upper_bound.py [n ...]      (default: 100 = as generated, 24, 4)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mgnns_amd import harness, synth  # noqa: E402

cuts = [int(a) for a in sys.argv[1:]] or [100, 24, 4]
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS["mvsa_multiple_b256"]
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=256, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision("bf16").set_attention("faithful")
calls = {}
for n in cuts:
    v = dict(inp)
    tok = inp["text"].copy()
    tok[:, n:] = 0
    lens = np.minimum(inp["text_lens"], n)
    v.update(text=tok, text_lens=lens, text_mask=(tok != 0).astype(np.float32))
    calls[n] = harness.call_args(v, dev)
    print("cut %3d: mean length %.1f, longest %d" % (n, float(lens.mean()), int(lens.max())), flush=True)
with torch.no_grad():
    for r in range(3):
        for n in cuts:
            res = bench.graphed_variant(model, calls[n], 256, 30, 5, "", in_flight=2)
            one = res.get("serial_replay", res)
            print("round %d, documents cut to %3d tokens: %.4f ms two in flight, %.4f ms one at a time" % (r + 1, n, res["ms_per_step"], one["ms_per_step"]), flush=True)
