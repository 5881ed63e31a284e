#!/usr/bin/env python
"""Per-phase cycle sums of imgbank_pool_bf16 (library built with MGNNS_HIPCC_FLAGS=-DMG_IMG_TRACE): time a wave spends
waiting for / converting the next map slice, in the MFMAs, and in the LDS write + barrier, over the 16 slices."""
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
for i, name in enumerate(["wg0 consumer0", "wg0 producer0", "wg301 consumer0", "wg301 producer0"]):
    t = buf[i * 8:(i + 1) * 8]
    if i % 2 == 0:
        print("%s: work (A reads + MFMAs + W requests) %d  barrier wait %d   [s_memtime ticks, 16 slices]" % (name, t[0], t[1]))
    else:
        print("%s: wait for the set's first row %d  emit (rest of the waits + pool + cvt + LDS write) %d  refill issue %d  barrier wait %d"
              % (name, t[0], t[1], t[2], t[3]))
