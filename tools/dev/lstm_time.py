#!/usr/bin/env python
"""BiLSTM text bank alone (bf16 mode, B=256 of the bench workload): python tools/dev/lstm_time.py   [MGNNS_LIB=... variants]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import harness, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.CONFIGS["mvsa_multiple_b256"]
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=256, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision("bf16")
text, lens = torch.as_tensor(inp["text"]).to(dev), torch.as_tensor(inp["text_lens"]).to(dev)
with torch.no_grad():
    for B in (256, 32):
        t, l = text[:B].contiguous(), lens[:B].contiguous()
        for _ in range(5):
            model._text_bank(t, l)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            model._text_bank(t, l)
        b.record()
        torch.cuda.synchronize()
        print("B=%d text bank (prep + 2 x (projection GEMM + recurrence)): %.1f us" % (B, a.elapsed_time(b) / 20 * 1e3))
