import sys, torch, numpy as np
sys.path.insert(0, ".")
from mgnns_amd import ops
dev = "cuda:0"; B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
sp = lambda w: ops.pack_weight_bf16_split(w.contiguous())
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for C in (80, 365):
    NLQ = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    Q = rn(NLQ, 300); pooled = torch.relu(torch.randn(B, 2, 2048, device=dev, generator=g))
    packed = {"wk": sp(rn(300, C)), "bk": rn(300), "wv": sp(rn(300, C)), "bv": rn(300), "wc": sp(rn(100, 300)), "bc": rn(100), "n5": 100, "C": C,
              "xl": sp(rn(300, NLQ * 100)), "bxl": rn(300), "n_out": 300}
    Gp = sp(rn(C, 2048)); nq = (sp(rn(1024, 300)), rn(1024), 1024)
    for terms in (1, 3):
        print("C=%d terms=%d: %.1f us (+qh)  %.1f us (no qh)  | one workgroup per tile: %.1f us (+qh)" % (
            C, terms, t(lambda: ops.label_tail_bf16(pooled, Gp, Q, 5, packed, next_q=nq, terms=terms)),
            t(lambda: ops.label_tail_bf16(pooled, Gp, Q, 5, packed, terms=terms)),
            t(lambda: ops.label_tail_bf16(pooled, Gp, Q, 5, packed, next_q=nq, terms=terms, cluster=False))))

import ctypes, os
from mgnns_amd import _lib
if hasattr(_lib.lib(), "mgnns_debug_lt_trace"):
    C = 365
    packed["C"] = C
    for terms in (1, 3):
        ops.label_tail_bf16(pooled, Gp, Q, 5, packed, next_q=nq, terms=terms); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        fn = _lib.lib().mgnns_debug_lt_trace; fn.argtypes = [ctypes.c_void_p]; fn(ctypes.addressof(buf))
        t = list(buf)[:8]
        print("terms=%d phases (ticks): init %d, readout %d, x-convert %d, KV %d, label-loop %d, x_linear %d, store+qh %d, total %d" %
              (terms, t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[5]-t[4], t[6]-t[5], t[7]-t[6], t[7]-t[0]))
