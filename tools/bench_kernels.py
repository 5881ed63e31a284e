#!/usr/bin/env python
"""Micro-benchmarks of individual kernels at the headline shapes (B=256): time per launch by HIP events,
algorithmic TFLOP/s / GB/s.  Usage: python tools/bench_kernels.py [mha_bf16] [mha_f32] [imgbank] [lstm] [textgcn]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import ops, synth  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def mha(kind, B=256, L=196, H=8, masked=False):
    g = torch.Generator(device=DEV).manual_seed(0)
    bank = torch.randn(B, L, 300, device=DEV, generator=g)
    qh = torch.randn(B, H * 128, device=DEV, generator=g)
    wk = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
    wv = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
    bk = torch.randn(H * 128, device=DEV, generator=g) * 0.05
    bv = torch.randn(H * 128, device=DEV, generator=g) * 0.05
    mask = None
    if masked:
        rs = np.random.RandomState(0)
        lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, B))), 4, L).astype(int)
        lens[0] = L
        mask = torch.zeros(B, L, device=DEV)
        for b in range(B):
            mask[b, :lens[b]] = 1
    fl = B * (4.0 * L * 300 * H * 128 + 4.0 * H * 128 * L)
    if kind == "bf16":
        bb = ops.cast_pad_bf16(bank)
        wp = ops.pack_kv_weights_bf16(wk, wv, H, 128)
        ms = timeit(lambda: ops.sq_mha_core_bf16(qh, bb, mask, H, 128, wp, bk, bv))
    else:
        ms = timeit(lambda: ops.sq_mha_core(qh, bank, mask, H, 128, wk, bk, wv, bv))
    print("sq_mha_core_%s B=%d L=%d H=%d masked=%s: %.1f us  %.1f TFLOP/s (full-L algorithmic)"
          % (kind, B, L, H, masked, ms * 1e3, fl / ms / 1e9))


def imgbank(B=256):
    g = torch.Generator(device=DEV).manual_seed(0)
    feat = torch.relu(torch.randn(B, 2048, 196, device=DEV, generator=g))
    w = torch.randn(300, 2048, device=DEV, generator=g) * 0.05
    bias = torch.randn(300, device=DEV, generator=g) * 0.05
    wt = ops.transpose_pad(w, ops.IMGBANK_LDW)
    ms = timeit(lambda: ops.imgbank_pool(feat, wt, bias, 300))
    fl = B * 2.0 * 196 * 2048 * 300
    by = B * 2048 * 196 * 4.0
    print("imgbank_pool f32 B=%d: %.1f us  %.1f TFLOP/s  %.0f GB/s (map read)" % (B, ms * 1e3, fl / ms / 1e9, by / ms / 1e6))
    if hasattr(ops, "imgbank_pool_bf16"):
        wp = ops.pack_imgbank_weights_bf16(w)
        ms = timeit(lambda: ops.imgbank_pool_bf16(feat, wp, bias, 300))
        print("imgbank_pool bf16 B=%d: %.1f us  %.1f TFLOP/s  %.0f GB/s (map read)" % (B, ms * 1e3, fl / ms / 1e9, by / ms / 1e6))


if __name__ == "__main__":
    what = sys.argv[1:] or ["mha_bf16", "mha_f32", "imgbank"]
    if "mha_bf16" in what:
        mha("bf16")
        mha("bf16", L=100, masked=True)
    if "mha_f32" in what:
        mha("f32")
        mha("f32", L=100, masked=True)
    if "imgbank" in what:
        imgbank()
