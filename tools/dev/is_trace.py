#!/usr/bin/env python
"""Phase sums (s_memtime ticks over the 32 slices) of imgbank_split_kernel, waves 0 / 4 of workgroup 0.
Build: python tools/dev/build_variant.py is_trace imgbank_split.hip -DMG_IS_TRACE ; run with MGNNS_LIB=mgnns_amd/variants/lib_is_trace.so"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=DEV).manual_seed(0)
feat = torch.relu(torch.randn(B, 2048, 196, device=DEV, generator=g))
w = torch.randn(300, 2048, device=DEV, generator=g) * 0.05
bias = torch.randn(300, device=DEV, generator=g) * 0.05
ws = ops.pack_weight_bf16_split(w)
for _ in range(3):
    ops.imgbank_pool_split(feat, ws, bias, 300, want_f32=False, want_split=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    ops.imgbank_pool_split(feat, ws, bias, 300, want_f32=False, want_split=True)
b.record()
torch.cuda.synchronize()
print("imgbank_pool_split B=%d: %.1f us per launch (eager)" % (B, a.elapsed_time(b) / 10 * 1e3))
buf = (ctypes.c_ulonglong * 16)()
fn = _lib.lib().mgnns_debug_is_trace
fn.argtypes = [ctypes.c_void_p]
fn(ctypes.addressof(buf))
names = ["convert (head)", "pooled maxima", "W requests + wait", "MFMA blocks", "convert (tail)", "barrier"]
for wv in range(2):
    t = list(buf)[8 * wv: 8 * wv + 6]
    print("wave %d: " % (4 * wv) + ", ".join("%s %d" % (n, v) for n, v in zip(names, t)) + ", total %d ticks" % sum(t))
