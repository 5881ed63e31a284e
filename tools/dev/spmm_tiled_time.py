#!/usr/bin/env python
"""Cache-cold time of the LDS-tiled bf16 SpMM at configs[4]'s density 1e-2 (N = 10 000, F = 1024 / 2048; hipGraph replay of one rotation over
> 640 MiB of operand sets).  MGNNS_LIB=<variant> measures an ablation build (e.g. -DMG_SPMM_ABLATE_FIXED: results wrong on purpose)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mgnns_amd import ops, stress
dev = "cuda:0"; n = 10000
g = torch.Generator(device=dev).manual_seed(0)
csr_np = stress.random_csr(n, 1e-2, 1)
adj = ops.SparseAdjBf16(stress.csr_to_device(csr_np, dev))
nnz = csr_np[1].size
for F in (1024, 2048):
    k = stress._sets_for(2.0 * n * F * 2)
    xs = [torch.randn(n, F, device=dev, generator=g).bfloat16() for _ in range(k)]
    ys = [torch.empty_like(x) for x in xs]
    by = nnz * 6.0 + 2.0 * n * F * 2
    for geo in (None, (8, 10, 128), (8, 20, 128), (4, 20, 256), (4, 10, 256)):
        run = lambda x, y: ops.spmm_bf16(adj, x, act=ops.ACT_LRELU2, out=y, path="tiled", geometry=geo)
        try:
            ms = min(stress.time_cold(run, list(zip(xs, ys))) for _ in range(3))
        except Exception as e:
            print(json.dumps({"geo": geo, "F": F, "error": str(e)[:100]}))
            continue
        rb = -(-n // (16 * (geo[1] if geo else 10)))
        print(json.dumps({"tiled_d1e-2": 1, "F": F, "geometry(lane_bytes, rows_per_wave, tile_cols)": geo, "us": round(ms * 1e3, 1),
                          "GBps": round(by / ms / 1e6), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 4),
                          "staged_GB": round(rb * n * F * 2 / 1e9, 2), "lib": os.environ.get("MGNNS_LIB", "default")}), flush=True)
    del xs, ys
