#!/usr/bin/env python
"""Phase cycles of mha_tail_bf16 (wave 0 of workgroup 0; library built with -DMG_TAIL_TRACE via tools/dev/build_variant.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
B, H = 256, 8
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *shape: torch.randn(*shape, device=DEV, generator=g) * 0.05
o, q = torch.randn(B, H * 128, device=DEV, generator=g), torch.randn(B, 300, device=DEV, generator=g)
fc, w1, w2, wq = r(300, H * 128), r(300, 300), r(300, 300), r(H * 128, 300)
common = {"fc_b": r(300), "g1": r(300) + 1, "be1": r(300), "b1": r(300), "b2": r(300), "g2": r(300) + 1, "be2": r(300)}
pkbf = dict(common, fc=ops.pack_weight_bf16_split(fc), w1=ops.pack_weight_bf16_split(w1), w2=ops.pack_weight_bf16_split(w2))
nxbf = (ops.pack_weight_bf16_split(wq), r(H * 128), H * 128)
for _ in range(5):
    ops.mha_tail_bf16(o, q, pkbf, 1e-6, nxbf, terms=1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
fn = _lib.lib().mgnns_debug_tail_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.addressof(buf)) == 0
names = ["params + ring prime -> o staged", "fc GEMM", "LN1 + split", "w1 GEMM", "relu + store", "w2 GEMM", "LN2 + out + split", "wq GEMM"]
t = list(buf)
print("total %d ticks: " % (t[8] - t[0]) + "  ".join("%s %d" % (names[i], t[i + 1] - t[i]) for i in range(8)))
