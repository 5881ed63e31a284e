#!/usr/bin/env python
"""Phase timeline of sq_mha_core_bf16 from in-kernel s_memtime stamps (library built with
MGNNS_HIPCC_FLAGS=-DMG_MHA_TRACE).  Prints, for wave 0 / wave 4 of workgroups 0 and 129, the cycles spent in staging,
each K/V GEMM and each epilogue."""
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
names = ["entry", "dma issued", "staged"] + [x for hp in range(4) for x in ("K gemm %d" % hp, "K epi %d" % hp, "V gemm %d" % hp, "V epi %d" % hp)]
for w in range(4):
    t = list(buf[w * 64:(w + 1) * 64])
    print("workgroup %d wave %d: total %d ticks" % (0 if w < 2 else 129, 0 if w % 2 == 0 else 4, t[len(names) - 1] - t[0]))
    print("  " + "  ".join("%s %d" % (names[i], t[i] - t[i - 1]) for i in range(1, len(names))))
    print("  pair 1 K gemm k-steps: " + " ".join(str(t[32 + k + 1] - t[32 + k]) for k in range(9)) + "  (from epilogue end to k0: %d)" % (t[32] - t[6]))
    print("  pair 1 V gemm k-steps: " + " ".join(str(t[44 + k + 1] - t[44 + k]) for k in range(9)) + "  (from epilogue end to k0: %d)" % (t[44] - t[8]))
base = buf[2 * 64 + 0]
for w in (2, 3):
    t = list(buf[w * 64:(w + 1) * 64])
    print("wg129 wave %d absolute (k cycles): " % (0 if w == 2 else 4) + " ".join("%.1f" % ((t[i] - base) / 1e3) for i in range(2, 19)))
