#!/usr/bin/env python
"""Phase timeline of sq_mha32_core_kernel from in-kernel s_memtime stamps (variant library built with -DMG_MHA32_TRACE:
python tools/dev/build_variant.py trace32 sq_mha32_bf16.hip -fno-slp-vectorize -DMG_MHA32_TRACE; MGNNS_LIB=mgnns_amd/variants/lib_trace32.so).
Wave 0 runs the K units of slice 0, wave 4 (same SIMD) its V units; workgroups 0 and 129."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
B, L, H = 256, 196, 8
g = torch.Generator(device=DEV).manual_seed(0)
bank = ops.cast_pad_bf16(torch.randn(B, L, 300, device=DEV, generator=g))
qh = torch.randn(B, H * 128, device=DEV, generator=g)
wk = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
wv = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
bk = torch.zeros(H * 128, device=DEV)
wp = ops.pack_kv_weights_bf16(wk, wv, H, 128, form=32)
for _ in range(5):
    ops.sq_mha_core_bf16(qh, bank, None, H, 128, wp, bk, bk, want_attn=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
fn = _lib.lib().mgnns_debug_mha32_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.addressof(buf)) == 0
if os.environ.get("MGNNS_TRACE_RAW") == "1":       # rows_body: stamps 3.. = [unit-0 GEMM] then per unit (GEMM+epilogue inside, finish)
    for w in range(4):
        t = list(buf[w * 64:(w + 1) * 64])
        d = [t[i + 1] - t[i] for i in range(2, 40) if t[i + 1] > t[i]]
        print("workgroup %d wave %d: DMA landed %d, staged %d; deltas: %s ; end at %d"
              % (0 if w < 2 else 129, 0 if w % 2 == 0 else 4, t[1] - t[0], t[2] - t[0], d, max(t) - t[0]))
    sys.exit(0)
for w in range(4):
    t = list(buf[w * 64:(w + 1) * 64])
    print("workgroup %d wave %d: own DMA landed %d, staged (barrier) %d ticks after entry"
          % (0 if w < 2 else 129, 0 if w % 2 == 0 else 4, t[1] - t[0], t[2] - t[0]))
    rows, prev = [], t[2]
    for u in range(2 * H):                      # the units this wave drew from its slice's queue (K and V mixed)
        g, e = t[3 + 2 * u], t[4 + 2 * u]
        if g <= prev or e < g:
            break
        rows.append("(gemm %d, epi %d)" % (g - prev, e - g))
        prev = e
    print("   %d units: " % len(rows) + "  ".join(rows))
    print("   end at %d ticks" % (prev - t[0]))
