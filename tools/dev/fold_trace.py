#!/usr/bin/env python
"""Phase cycles of folded_attn_bf16_kernel (wave 0 of workgroups 0 and 129; library built with -DMG_FOLD_TRACE via
tools/dev/build_variant.py fold_trace sq_mha_folded_bf16.hip -DMG_FOLD_TRACE, run with MGNNS_LIB=...)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
H = 8
for B, L, masked in ((256, 196, False), (256, 100, True), (64, 196, False)):
    g = torch.Generator(device=DEV).manual_seed(0)
    bank = ops.cast_pad_bf16(torch.randn(B, L, 300, device=DEV, generator=g))
    u = torch.randn(B, H * 300, device=DEV, generator=g) * 0.3
    mask = None
    if masked:
        mask = torch.zeros(B, L, device=DEV)
        for b in range(B):
            mask[b, :max(4, (b * 37) % L)] = 1
        mask[0, :] = 1
        mask[129 % B, :] = 1
    for _ in range(5):
        ops.sq_mha_folded_bf16(u, bank, mask, H, 128, want_attn=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    fn = _lib.lib().mgnns_debug_fold_trace
    fn.argtypes = [ctypes.c_void_p]
    assert fn(ctypes.addressof(buf)) == 0
    names = ["mask / live rows", "DMA issue", "U + P zero", "wait DMA + barrier", "GEMM 1", "softmax", "GEMM 2", "C store"]
    for w in range(2):
        t = list(buf)[16 * w:16 * w + 9]
        print("B=%d L=%d masked=%s wg %d: total %d cycles: " % (B, L, masked, 129 * w, t[8] - t[0]) +
              "  ".join("%s %d" % (names[i], t[i + 1] - t[i]) for i in range(8)))
