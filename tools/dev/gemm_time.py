#!/usr/bin/env python
"""Dense bf16 GEMM of configs[4] ([n, n] adjacency x [n, F]), cache-cold rotation, hipGraph replays: us and fraction of 2.5 PF.
    python tools/dev/gemm_time.py [F ...]      MGNNS_GEMM_TILE=128 selects the 256 x 128 kernel"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops  # noqa: E402

dev = "cuda:0"
n = 10000
kp = (n + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(0)
for F in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    adjs = [torch.zeros(n, kp, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    xts = [torch.zeros(F, kp, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    for t in adjs + xts:
        t[:, :n] = torch.randn(t.shape[0], n, device=dev, generator=g).bfloat16()
    out = torch.empty(n, F, device=dev)
    for i in range(3):
        ops.gemm_bf16_nt(adjs[i], xts[i], out=out)
    torch.cuda.synchronize()
    ds = []
    for r in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(6):
            ops.gemm_bf16_nt(adjs[i % 3], xts[i % 3], out=out)
        b.record()
        torch.cuda.synchronize()
        ds.append(a.elapsed_time(b) / 6 * 1e3)
    us = statistics.median(ds)
    print("F=%d: %.1f us (min %.1f)  %.1f %% of 2.5 PF" % (F, us, min(ds), 2.0 * n * n * F / (us * 1e-6) / 2.5e15 * 100), flush=True)
    if os.environ.get("MGNNS_GEMM_TRACE") == "1":
        import ctypes
        from mgnns_amd import _lib
        buf = (ctypes.c_ulonglong * 4)()
        fn = _lib.lib().mgnns_debug_gemm_trace
        fn.argtypes = [ctypes.c_void_p]
        assert fn(ctypes.addressof(buf)) == 0
        print("   s_memtime: workgroup 0: %d ticks for %d 32-wide slices (%.0f per slice); workgroup 100: %d for %d  => %.2f GHz if the launch is %.1f us"
              % (buf[0], buf[1], buf[0] / max(1, buf[1]), buf[2], buf[3], buf[0] / (us * 1e3), us))
    del adjs, xts
