#!/usr/bin/env python
"""Phase breakdown of the persistent implicit-GEMM convolution from in-kernel cycle counters (library built with
MGNNS_HIPCC_FLAGS=-DMG_CONV_TRACE): cycles wave 0 (issues its DMA pieces right after the slice barrier) and wave 4 (issues
them after its second k-step) of workgroup 0 spend per slice in each part of the loop."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
fn = _lib.lib().mgnns_debug_conv_trace
fn.argtypes = [ctypes.c_void_p]
NAMES = ["total", "DMA issue", "k0: reads k1 + wait + 16 MFMA", "slice barrier", "k1: reads k0' + 16 MFMA", "epilogue", "slices", "tiles", "tile loads"]


def run(k, cin, cout, hw, B=128, res=True):
    x = torch.randn(B, hw, hw, cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(cout, k * k * cin, device=DEV) * 0.05).to(torch.bfloat16)
    b = torch.randn(cout, device=DEV)
    r = torch.randn(B, hw, hw, cout, device=DEV).to(torch.bfloat16) if res else None
    for _ in range(3):
        ops.conv_bf16_nhwc(x, w, b, k, 1, k // 2, residual=r)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert fn(ctypes.addressof(buf)) == 0
    print("conv %dx%d %d->%d @%d^2 x%d res=%d" % (k, k, cin, cout, hw, B, res))
    for wv in range(2):
        t = list(buf[wv * 16:wv * 16 + 9])
        n = max(t[6], 1)
        print("  wave %d: %d slices, %d tiles, total %d cycles = %.0f per slice" % (4 * wv, t[6], t[7], t[0], t[0] / n))
        print("    per slice: " + ", ".join("%s %.0f" % (NAMES[i], t[i] / n) for i in (1, 2, 3, 4, 8)) + "; epilogue %.0f per tile" % (t[5] / max(t[7], 1)))


run(3, 256, 256, 28, res=False)
run(1, 1024, 256, 28, res=False)
run(1, 256, 1024, 28)
run(1, 64, 256, 112)
