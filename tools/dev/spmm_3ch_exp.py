#!/usr/bin/env python
"""configs[4] has three channels: does ONE launch over the block-diagonal union of the three adjacencies (30 000 rows, X / Y
stacked) beat three launches?  Cache-cold like stress.measure.  python tools/dev/spmm_3ch_exp.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops, stress  # noqa: E402

DEV = "cuda:0"
n = stress.N_NODES
g = torch.Generator(device=DEV).manual_seed(0)
for dens in (stress.DENSITIES[0],):
    csrs = [stress.random_csr(n, dens, 1 + 10 * c) for c in range(3)]
    adjs = [ops.SparseAdjBf16(stress.csr_to_device(c, DEV)) for c in csrs]
    adjs_u = [ops.SparseAdjBf16(stress.csr_to_device(c, DEV), sort_rows=False) for c in csrs]
    rp = np.concatenate([[0]] + [c[0][1:].astype(np.int64) + sum(int(x[0][-1]) for x in csrs[:i]) for i, c in enumerate(csrs)]).astype(np.int32)
    col = np.concatenate([c[1] + i * n for i, c in enumerate(csrs)]).astype(np.int32)
    val = np.concatenate([c[2] for c in csrs])
    big = ops.SparseAdjBf16(stress.csr_to_device((rp, col, val), DEV))
    big_u = ops.SparseAdjBf16(stress.csr_to_device((rp, col, val), DEV), sort_rows=False)
    for F in (1024, 2048):
        k = stress._sets_for(2.0 * 3 * n * F * 2)
        xs = [torch.randn(3 * n, F, device=DEV, generator=g).bfloat16() for _ in range(k)]
        ys = [torch.empty_like(x) for x in xs]
        by1 = adjs[0].nnz * 6.0 + 2.0 * n * F * 2
        by3 = big.nnz * 6.0 + 2.0 * 3 * n * F * 2

        def three(x, y):
            for c in range(3):
                ops.spmm_bf16(adjs[c], x[c * n:(c + 1) * n], act=ops.ACT_LRELU2, out=y[c * n:(c + 1) * n])
        ms3 = stress.time_cold(three, list(zip(xs, ys)))
        print("F=%d three launches, rows sorted by length: %.2f us per 3 channels = %.2f per channel, %.0f GB/s (%.1f %% of 8 TB/s)"
              % (F, ms3 * 1e3, ms3 * 1e3 / 3, by3 / ms3 / 1e6, by3 / ms3 / 1e6 / 80))

        def three_u(x, y):
            for c in range(3):
                ops.spmm_bf16(adjs_u[c], x[c * n:(c + 1) * n], act=ops.ACT_LRELU2, out=y[c * n:(c + 1) * n])
        ms3u = stress.time_cold(three_u, list(zip(xs, ys)))
        print("F=%d three launches, rows in graph order: %.2f us = %.2f per channel (%.1f %% of 8 TB/s)"
              % (F, ms3u * 1e3, ms3u * 1e3 / 3, by3 / ms3u / 1e6 / 80))
        msu = stress.time_cold(lambda x, y: ops.spmm_bf16(big_u, x, act=ops.ACT_LRELU2, out=y), list(zip(xs, ys)))
        print("F=%d ONE launch, block diagonal, rows in graph order: %.2f us = %.2f per channel (%.1f %% of 8 TB/s)"
              % (F, msu * 1e3, msu * 1e3 / 3, by3 / msu / 1e6 / 80))
        for name, variant in (("auto", 0), ("ring NSL=1 RI=8 wg256", (1 << 29) | (256 << 8) | 2), ("ring NSL=1 RI=8 wg384", (1 << 29) | (384 << 8) | 2),
                              ("ring NSL=1 RI=8 wg64", (1 << 29) | (64 << 8) | 2),
                              ("reg NS=2 RU=1 wg384", (1 << 30) | (384 << 4) | 4), ("reg NS=2 RU=2 wg384", (1 << 30) | (384 << 4) | 5),
                              ("reg NS=2 RU=1 wg768", (1 << 30) | (768 << 4) | 4)):
            try:
                ms = stress.time_cold(lambda x, y: ops.spmm_bf16(big, x, act=ops.ACT_LRELU2, out=y, variant=variant), list(zip(xs, ys)))
            except Exception as e:
                print("   %s: %s" % (name, e))
                continue
            print("F=%d ONE launch, block diagonal [%s]: %.2f us = %.2f per channel, %.0f GB/s (%.1f %% of 8 TB/s)"
                  % (F, name, ms * 1e3, ms * 1e3 / 3, by3 / ms / 1e6, by3 / ms / 1e6 / 80))
        msc = stress.time_cold(lambda d, s_: d.copy_(s_), list(zip(ys, xs)))
        print("F=%d copy of the stacked X -> Y: %.2f us (%.0f GB/s)" % (F, msc * 1e3, 2.0 * 3 * n * F * 2 / msc / 1e6))
        # check: one launch == three launches
        ya = torch.empty_like(xs[0]); yb = torch.empty_like(xs[0])
        three(xs[0], ya); ops.spmm_bf16(big, xs[0], act=ops.ACT_LRELU2, out=yb); yc = torch.empty_like(xs[0]); three_u(xs[0], yc)
        torch.cuda.synchronize()
        print("   equal:", bool(torch.equal(ya, yb)), bool(torch.equal(ya, yc)))
        del xs, ys
