#!/usr/bin/env python
"""sq_mha_core_bf16 across batch sizes / lengths (one line each): python tools/dev/mha_sizes.py  [MGNNS_LIB=...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B in (256, 128, 64, 32):
    for L, masked in ((196, False), (100, True)):
        H = 8
        g = torch.Generator(device=DEV).manual_seed(0)
        bank = ops.cast_pad_bf16(torch.randn(B, L, 300, device=DEV, generator=g))
        qh = torch.randn(B, H * 128, device=DEV, generator=g)
        wk = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
        wv = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
        bk = torch.zeros(H * 128, device=DEV)
        wp = ops.pack_kv_weights_bf16(wk, wv, H, 128)
        mask = None
        if masked:
            rs = np.random.RandomState(0)
            lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, B))), 4, L).astype(int)
            lens[0] = L
            mask = torch.zeros(B, L, device=DEV)
            for b in range(B):
                mask[b, :lens[b]] = 1
        us = timeit(lambda: ops.sq_mha_core_bf16(qh, bank, mask, H, 128, wp, bk, bk, want_attn=False))
        print("B=%3d L=%3d masked=%d: %.1f us" % (B, L, masked, us))
