#!/usr/bin/env python
"""imgbank_pool_bf16 at B = 256 on rotating maps (3 x 411 MB: nothing of a map is left in the 256-MB Infinity Cache when its turn
comes again); median / min of 12 replays of a hipGraph of 12 launches.  MGNNS_LIB selects the library (A/B of kernel variants on one box)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=DEV).manual_seed(0)
feats = [torch.relu(torch.randn(B, 2048, 196, device=DEV, generator=g)) for _ in range(3)]
w = torch.randn(300, 2048, device=DEV, generator=g) * 0.05
bias = torch.randn(300, device=DEV, generator=g) * 0.05
wp = ops.pack_imgbank_weights_bf16(w)
for f in feats:
    ops.imgbank_pool_bf16(f, wp, bias, 300)
torch.cuda.synchronize()
st = torch.cuda.Stream()
gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    with torch.cuda.graph(gr, stream=st):
        for i in range(12):
            ops.imgbank_pool_bf16(feats[i % 3], wp, bias, 300)
    gr.replay()
    torch.cuda.synchronize()
    rounds = []
    for r in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gr.replay()
        b.record()
        torch.cuda.synchronize()
        rounds.append(a.elapsed_time(b) / 12 * 1e3)
by = B * 2048 * 196 * 4.0
med = statistics.median(rounds)
print("imgbank_pool_bf16 B=%d: median %.1f us (min %.1f, max %.1f)  %.2f TB/s of map" % (B, med, min(rounds), max(rounds), by / med / 1e6))
