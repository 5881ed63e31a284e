#!/usr/bin/env python
"""Per-phase cycle sums of imgbank_pool_bf16 (library built with MGNNS_HIPCC_FLAGS=-DMG_IMG_TRACE): time a wave spends
issuing the DMA rows, converting a k-step, in its MFMAs, waiting for its own rows and in the barrier, over the 64 k-steps."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
B = 256
g = torch.Generator(device=DEV).manual_seed(0)
feat = torch.relu(torch.randn(B, 2048, 196, device=DEV, generator=g))
w = torch.randn(300, 2048, device=DEV, generator=g) * 0.05
bias = torch.randn(300, device=DEV, generator=g) * 0.05
wp = ops.pack_imgbank_weights_bf16(w)
for _ in range(3):
    ops.imgbank_pool_bf16(feat, wp, bias, 300)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
fn = _lib.lib().mgnns_debug_img_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.addressof(buf)) == 0
for i, name in enumerate(["wg0 wave0 (MFMAs first)", "wg0 wave4 (conversion first)", "wg129 wave0", "wg129 wave4"]):
    t = buf[i * 8:(i + 1) * 8]
    print("%s: DMA issue %d  convert + pool %d  MFMAs + W requests %d  wait for own rows %d  barrier %d   [s_memtime ticks, 64 k-steps]"
          % (name, t[0], t[1], t[2], t[3], t[4]))
