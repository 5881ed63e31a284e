#!/usr/bin/env python
"""Phase cycles of mha_tail_c16 (wave 0 of workgroup 0 = rank 0 of tile 0; library built with -DMG_TAIL_TRACE via
tools/dev/build_variant.py tailtrace mha_tail.hip -DMG_TAIL_TRACE, run with MGNNS_LIB=...)."""
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
c, q = torch.randn(B, H * 300, device=DEV, generator=g).to(torch.bfloat16), torch.randn(B, 300, device=DEV, generator=g)
fc, w1, w2, wq = r(300, H * 300), r(300, 300), r(300, 300), r(H * 300, 300)
pk = {"fc_b": r(300), "g1": r(300) + 1, "be1": r(300), "b1": r(300), "b2": r(300), "g2": r(300) + 1, "be2": r(300),
      "fc": ops.pack_weight_bf16_split(fc), "w1": ops.pack_weight_bf16_split(w1), "w2": ops.pack_weight_bf16_split(w2)}
nx = (ops.pack_weight_bf16_split(wq), r(H * 300), H * 300)
fn = _lib.lib().mgnns_debug_tail_trace
fn.argtypes = [ctypes.c_void_p]
names = ["params + ring prime -> c slice staged", "first product (K slice) + exchange", "LN1", "w1 GEMM", "relu + store", "w2 GEMM",
         "LN2 + out", "next composed query map"]
for cluster, ksplit in ((4, True), (2, True), (8, True), (4, False)):
    for _ in range(5):
        ops.mha_tail_c16(c, q, pk, 1e-6, nx, cluster=cluster, ksplit=ksplit)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert fn(ctypes.addressof(buf)) == 0
    t = list(buf)
    print("cluster %d ksplit %s: total %d cycles: " % (cluster, ksplit, t[8] - t[0]) + "  ".join("%s %d" % (names[i], t[i + 1] - t[i]) for i in range(8)))
