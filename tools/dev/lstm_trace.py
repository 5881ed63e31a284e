#!/usr/bin/env python
"""Phases of a recurrence step (library built with -DMG_LSTM_TRACE, MGNNS_LIB=...): s_memtime sums over the steps of the longest
chain of workgroup 0, waves 0 and 9."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import _lib, harness, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.CONFIGS["mvsa_multiple_b256"]
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=256, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision("bf16")
text, lens = torch.as_tensor(inp["text"]).to(dev), torch.as_tensor(inp["text_lens"]).to(dev)
with torch.no_grad():
    for _ in range(3):
        model._text_bank(text, lens)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
fn = _lib.lib().mgnns_debug_lstm_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.addressof(buf)) == 0
for i, name in enumerate(["wave 0", "wave 9"]):
    t = buf[i * 8:(i + 1) * 8]
    n = max(1, t[4])
    print("%s, %d steps: h reads + MFMAs %.0f  wait gx %.0f  activations + h write %.0f  barrier %.0f  = %.0f ticks per step"
          % (name, n, t[0] / n, t[1] / n, t[2] / n, t[3] / n, sum(t[:4]) / n))
