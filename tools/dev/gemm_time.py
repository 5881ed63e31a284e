#!/usr/bin/env python
"""Dense bf16 GEMM of configs[4] ([n, n] adjacency x [n, F]), cache-cold rotation, hipGraph replays: us and fraction of 2.5 PF.
    python tools/dev/gemm_time.py [F ...]      MGNNS_GEMM_TILE=128 selects the 256 x 128 kernel; every F is timed with the 160 x 256
    kernel off / forced / chosen by the launcher's estimate (ops.gemm_bf16_set_form 0 / 1 / 2)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
# arguments: F (the adjacency product 10 000 x F x 10 000) or M,N,K (any product, e.g. 10000,2048,1024 = X1 . W2 of configs[4])
shapes = [(10000, int(a), 10000) if "," not in a else tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(10000, 1024, 10000), (10000, 2048, 10000)]
for n, F, kk in shapes:
    kp = (kk + 63) // 64 * 64
    sets = max(3, int(700e6 / (2.0 * kp * (n + F))) + 1)              # > 640 MB of operands in rotation: cache-cold
    adjs = [torch.zeros(n, kp, device=dev, dtype=torch.bfloat16) for _ in range(sets)]
    xts = [torch.zeros(F, kp, device=dev, dtype=torch.bfloat16) for _ in range(sets)]
    for t in adjs + xts:
        t[:, :kk] = torch.randn(t.shape[0], kk, device=dev, generator=g).bfloat16()
    out = torch.empty(n, F, device=dev)
    outs = {}
    for form in (0, 1, 3, 2, 0, 1, 3):
        ops.gemm_bf16_set_form(form)
        for i in range(3):
            ops.gemm_bf16_nt(adjs[i % sets], xts[i % sets], out=out)
        ops.gemm_bf16_nt(adjs[0], xts[0], out=out)
        torch.cuda.synchronize()
        outs.setdefault(form, out.clone())
        ds = []
        for r in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            reps = max(6, sets)
            for i in range(reps):
                ops.gemm_bf16_nt(adjs[i % sets], xts[i % sets], out=out)
            b.record()
            torch.cuda.synchronize()
            ds.append(a.elapsed_time(b) / reps * 1e3)
        us = statistics.median(ds)
        print("%d x %d x %d, %s: %.1f us (min %.1f)  %.1f %% of 2.5 PF" % (n, F, kk, ("round-4 kernels", "160 x 256", "by estimate", "320 x 256")[form], us, min(ds),
                                                                                           2.0 * n * kk * F / (us * 1e-6) / 2.5e15 * 100), flush=True)
    ops.gemm_bf16_set_form(-1)
    for f_ in (1, 3):
        print("   max |%s - round-4 kernels| = %.3e of max |C| = %.3e" % (("", "160 x 256", "", "320 x 256")[f_], float((outs[0] - outs[f_]).abs().max()),
                                                                       float(outs[0].abs().max())))
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
