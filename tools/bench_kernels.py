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
    """Time per call.  MGNNS_BENCH_GRAPH=1: n calls captured in one hipGraph and replayed (no host launch cost)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("MGNNS_BENCH_GRAPH") == "1":
        st = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            fn()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(n):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                gr.replay()
            b.record()
            torch.cuda.synchronize()
        return a.elapsed_time(b) / (5 * n)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def mha(kind, B=256, L=196, H=8, masked=False):
    L = int(os.environ.get("MGNNS_BENCH_L", L)) if not masked else L          # (L = 192 with the 12-tile ablation build: the 13th tile's cost)
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
        plan = ops.sq_mha_plan(mask) if (masked and ops.MHA_CORE == 32 and ops.MHA_PACKED) else None
        if plan is not None:
            print("  plan: %d groups for %d samples, %d live rows" % (int(plan[0]), B, int(mask.sum())))
            print("  sq_mha_plan: %.1f us" % (timeit(lambda: ops.sq_mha_plan(mask)) * 1e3))
        ms = timeit(lambda: ops.sq_mha_core_bf16(qh, bb, mask, H, 128, wp, bk, bv, want_attn=False, plan=plan))
        print("  [core form %d%s]" % (ops.MHA_CORE, ", packed" if plan is not None else ""))
    elif kind == "split":
        sp = ops.split_pad_bf16(bank)
        wp = ops.pack_kv_weights_split(wk, wv, H, 128)
        print("  split_pad_bf16 (fp32 bank -> hi + lo images): %.1f us" % (timeit(lambda: ops.split_pad_bf16(bank)) * 1e3))
        plan = ops.sq_mha_split_plan(mask) if (masked and L <= ops.SPLIT_PLAN_MAX_L and os.environ.get("MGNNS_SPLIT_GROUPED", "1") == "1") else None
        if plan is not None:
            print("  group plan: %d groups for %d samples, %d live rows; sq_mha_split_plan %.1f us"
                  % (int(plan[0]), B, int(mask.sum()), timeit(lambda: ops.sq_mha_split_plan(mask)) * 1e3))
        ms = timeit(lambda: ops.sq_mha_core_split(qh, sp, mask, H, 128, wp, bk, bv, want_attn=False, plan=plan))
        print("  [3 MFMAs per product: %.1f TFLOP/s executed]" % (3 * fl / ms / 1e9))
    elif kind == "folded_c16":
        bb = ops.cast_pad_bf16(bank)
        u = torch.randn(B, H * 300, device=DEV, generator=g) * 0.3
        ms = timeit(lambda: ops.sq_mha_folded_bf16(u, bb, mask, H, 128, want_attn=False))
        by = bb.numel() * 2
        print("sq_mha_folded_bf16 B=%d L=%d H=%d masked=%s: %.1f us  %.0f GB/s (one bank read, %.1f MB)"
              % (B, L, H, masked, ms * 1e3, by / ms / 1e6, by / 1e6))
        return
    elif kind in ("folded", "folded_bf16"):
        x = ops.cast_pad_bf16(bank) if kind == "folded_bf16" else bank
        ms = timeit(lambda: ops.sq_mha_folded(qh, x, mask, H, 128, wk, wv, bv, want_attn=False))
        by = x.numel() * x.element_size()
        print("sq_mha_folded (%s bank) B=%d L=%d H=%d masked=%s: %.1f us (3 launches)  %.0f GB/s (one bank read)"
              % (kind, B, L, H, masked, ms * 1e3, by / ms / 1e6))
        return
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
    ws = ops.pack_weight_bf16_split(w)
    ms = timeit(lambda: ops.imgbank_pool_split(feat, ws, bias, 300))
    print("imgbank_pool split (bf16x3) B=%d, fp32 bank: %.1f us  %.1f TFLOP/s executed (3 MFMAs per product)" % (B, ms * 1e3, 3 * fl / ms / 1e9))
    ms = timeit(lambda: ops.imgbank_pool_split(feat, ws, bias, 300, want_f32=False, want_split=True))
    print("imgbank_pool split (bf16x3) B=%d, hi + lo images out: %.1f us" % (B, ms * 1e3))


def tail(B=256, H=8):
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *shape: torch.randn(*shape, device=DEV, generator=g) * 0.05
    o, q = torch.randn(B, H * 128, device=DEV, generator=g), torch.randn(B, 300, device=DEV, generator=g)
    fc, w1, w2, wq = r(300, H * 128), r(300, 300), r(300, 300), r(H * 128, 300)
    common = {"fc_b": r(300), "g1": r(300) + 1, "be1": r(300), "b1": r(300), "b2": r(300), "g2": r(300) + 1, "be2": r(300)}
    bq = r(H * 128)
    pk32 = dict(common, fc_wp=ops.pack_weight_f32(fc), w1_wp=ops.pack_weight_f32(w1), w2_wp=ops.pack_weight_f32(w2))
    nx32 = (ops.pack_weight_f32(wq), bq, H * 128)
    print("mha_tail fp32 (+next wq): %.1f us" % (timeit(lambda: ops.mha_tail(o, q, pk32, 1e-6, nx32)) * 1e3))
    print("mha_tail fp32 (last layer): %.1f us" % (timeit(lambda: ops.mha_tail(o, q, pk32, 1e-6, None)) * 1e3))
    pkbf = dict(common, fc=ops.pack_weight_bf16_split(fc), w1=ops.pack_weight_bf16_split(w1), w2=ops.pack_weight_bf16_split(w2))
    nxbf = (ops.pack_weight_bf16_split(wq), bq, H * 128)
    for terms in (1, 3):
        for ks in ((True, False) if terms == 1 else (False,)):
            tag = "terms=%d%s" % (terms, ", K of fc split over the cluster" if ks else "")
            print("mha_tail bf16 %s (+next wq): %.1f us" % (tag, timeit(lambda: ops.mha_tail_bf16(o, q, pkbf, 1e-6, nxbf, terms=terms, ksplit=ks)) * 1e3))
            print("mha_tail bf16 %s (last layer): %.1f us" % (tag, timeit(lambda: ops.mha_tail_bf16(o, q, pkbf, 1e-6, None, terms=terms, ksplit=ks)) * 1e3))
    for cl in (2, 8):
        print("mha_tail bf16 terms=1, K split, cluster %d (+next wq): %.1f us" % (cl, timeit(lambda: ops.mha_tail_bf16(o, q, pkbf, 1e-6, nxbf, terms=1, cluster=cl)) * 1e3))


def tail_c16(B=256, H=8):
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *shape: torch.randn(*shape, device=DEV, generator=g) * 0.05
    c, q = torch.randn(B, H * 300, device=DEV, generator=g).to(torch.bfloat16), torch.randn(B, 300, device=DEV, generator=g)
    fc, w1, w2, wq = r(300, H * 300), r(300, 300), r(300, 300), r(H * 300, 300)
    pk = {"fc_b": r(300), "g1": r(300) + 1, "be1": r(300), "b1": r(300), "b2": r(300), "g2": r(300) + 1, "be2": r(300),
          "fc": ops.pack_weight_bf16_split(fc), "w1": ops.pack_weight_bf16_split(w1), "w2": ops.pack_weight_bf16_split(w2)}
    nx = (ops.pack_weight_bf16_split(wq), r(H * 300), H * 300)
    for cluster, ksplit in ((0, True), (2, True), (4, True), (8, True), (2, False), (4, False)):
        print("mha_tail_c16 B=%d H=%d cluster=%d ksplit=%s: %.1f us (+next composed query map)  %.1f us (last layer)" % (
            B, H, cluster, ksplit, timeit(lambda: ops.mha_tail_c16(c, q, pk, 1e-6, nx, cluster=cluster, ksplit=ksplit)) * 1e3,
            timeit(lambda: ops.mha_tail_c16(c, q, pk, 1e-6, None, cluster=cluster, ksplit=ksplit)) * 1e3))


if __name__ == "__main__":
    what = sys.argv[1:] or ["mha_bf16", "mha_f32", "imgbank"]
    if "mha_bf16" in what:
        mha("bf16")
        mha("bf16", L=100, masked=True)
    if "mha_split" in what:
        mha("split")
        mha("split", L=100, masked=True)
        mha("split", B=32)
        mha("split", B=32, L=100, masked=True)
    if "mha_f32" in what:
        mha("f32")
        mha("f32", L=100, masked=True)
    if "mha_folded" in what:
        for k in ("folded", "folded_bf16"):
            mha(k)
            mha(k, L=100, masked=True)
            mha(k, H=1)
    if "folded_c16" in what:
        mha("folded_c16")
        mha("folded_c16", L=100, masked=True)
        mha("folded_c16", B=128)
        mha("folded_c16", B=512)
        tail_c16()
        tail_c16(B=64)
    if "imgbank" in what:
        imgbank()
    if "tail" in what:
        tail()
    if "tail_h" in what:
        for h in (1, 2, 4, 8):
            print("H =", h)
            tail(H=h)
