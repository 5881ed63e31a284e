#!/usr/bin/env python
"""The dense adjacency product exactly as bench.py's stress leg times it (stress.time_cold: 4 operand sets in rotation, one rotation
captured as a hipGraph, best of three): python tools/dev/gemm_cold.py [F ...]     MGNNS_LIB=<variant> for an A/B"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops, stress  # noqa: E402

dev = "cuda:0"
n = 10000
kp = (n + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(0)
for F in [int(a) for a in sys.argv[1:]] or [1024]:
    As = [ops.cast_pad_bf16(torch.rand(n, n, device=dev, generator=g) * (2.0 / n), ld=kp) for _ in range(4)]
    Sts = [ops.transpose_cast_bf16(torch.randn(n, F, device=dev, generator=g)) for _ in range(4)]
    Cs = [torch.empty(n, F, device=dev) for _ in range(4)]
    for form in (0, 2, 0, 2):
        ops.gemm_bf16_set_form(form)
        ms = stress.time_cold(lambda a, s, c: ops.gemm_bf16_nt(a, s, None, ops.ACT_LRELU2, out=c), list(zip(As, Sts, Cs)))
        print("F=%d, 160 x 256 kernel %s: %.1f us = %.1f %% of 2.5 PF" % (F, "off" if form == 0 else "by estimate", ms * 1e3,
                                                                          2.0 * n * n * F / ms / 1e9 / 2500.0 * 100), flush=True)
    ops.gemm_bf16_set_form(-1)
    del As, Sts, Cs
