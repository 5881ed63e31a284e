#!/usr/bin/env python
"""Phase timeline of sq_mha_core_bf16 from in-kernel s_memtime stamps (library built with
MGNNS_HIPCC_FLAGS=-DMG_MHA_TRACE).  Prints, for wave 0 / wave 4 of workgroups 0 and 129, the cycles spent in staging,
each unit's GEMM and epilogue (unit queue: csrc/sq_mha_bf16.hip)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
B, L, H = 256, 196, 8
g = torch.Generator(device=DEV).manual_seed(0)
bank = ops.cast_pad_bf16(torch.randn(B, L, 300, device=DEV, generator=g))
qh = torch.randn(B, H * 128, device=DEV, generator=g)
wk = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
wv = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
bk = torch.zeros(H * 128, device=DEV)
wp = ops.pack_kv_weights_bf16(wk, wv, H, 128)
for _ in range(5):
    ops.sq_mha_core_bf16(qh, bank, None, H, 128, wp, bk, bk, want_attn=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
fn = _lib.lib().mgnns_debug_mha_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.addressof(buf)) == 0
# stamps: 0 entry, 2 staged, then (GEMM end, epilogue end) per unit this wave drew from its slice's queue (K and V units of
# the heads in ticket order, shared with the other wave of the SIMD: waves 0 and 4 are such a pair)
NU = 8
for w in range(4):
    t = list(buf[w * 64:(w + 1) * 64])
    units = []
    for u in range(NU):
        g, e = t[3 + 2 * u], t[4 + 2 * u]
        prev = t[2 + 2 * u] if u else t[2]
        if g <= prev or e < g:
            break
        units.append((g - prev, e - g))
    end = t[2 + 2 * len(units)] if units else t[2]
    print("workgroup %d wave %d: staged after %d ticks, %d units recorded, last stamp at %d"
          % (0 if w < 2 else 129, 0 if w % 2 == 0 else 4, t[2] - t[0], len(units), end - t[0]))
    print("  (GEMM, epilogue) per unit: " + "  ".join("(%d, %d)" % u for u in units))
