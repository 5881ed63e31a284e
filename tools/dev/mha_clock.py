#!/usr/bin/env python
"""Effective shader clock of sq_mha_core_bf16 and its dependence on the DATA (VERDICT r5 item 3 / NOTES_r06 2b).
  * default library: wall time per launch (HIP events around 40 launches captured back to back in one hipGraph) on random operands, on
    zero operands, and on a bank / weights that hold one constant -- same instructions, different toggling;
  * with MGNNS_LIB=<a -DMG_MHA_TRACE build> (tools/dev/build_variant.py trace sq_mha_bf16.hip -DMG_MHA_TRACE): cycles from kernel
    entry to the last stamp of wave 0 / 4 of workgroups 0 and 129 (s_memtime = shader cycles) over the wall time of that launch.
    python tools/dev/mha_clock.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
B, L, H = 256, 196, 8
g = torch.Generator(device=DEV).manual_seed(0)


def operands(kind):
    if kind == "random":
        bank = torch.randn(B, L, 300, device=DEV, generator=g)
        wk, wv = (torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05 for _ in range(2))
        qh = torch.randn(B, H * 128, device=DEV, generator=g)
    elif kind == "zeros":
        bank, wk, wv, qh = (torch.zeros(s, device=DEV) for s in ((B, L, 300), (H * 128, 300), (H * 128, 300), (B, H * 128)))
    else:
        bank, wk, wv, qh = (torch.full(s, 0.5, device=DEV) for s in ((B, L, 300), (H * 128, 300), (H * 128, 300), (B, H * 128)))
    bk = torch.zeros(H * 128, device=DEV)
    return qh, ops.cast_pad_bf16(bank), ops.pack_kv_weights_bf16(wk, wv, H, 128), bk


def wall_us(args, n=40, reps=5):
    qh, bank, wp, bk = args
    fn = lambda: ops.sq_mha_core_bf16(qh, bank, None, H, 128, wp, bk, bk, want_attn=False)
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(3):
            fn()
        st.synchronize()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                fn()
        gr.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            gr.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


sets = {k: operands(k) for k in ("random", "zeros", "constant 0.5")}
res = {}
for rnd in range(3):                       # alternating: one box, one process
    for k, a in sets.items():
        res.setdefault(k, []).append(round(wall_us(a), 2))
for k, v in res.items():
    print("%-13s operands: %s us per launch (40 back to back in one graph, three alternating rounds)" % (k, v))
try:
    fn = _lib.lib().mgnns_debug_mha_trace
except AttributeError:
    sys.exit(0)
fn.argtypes = [ctypes.c_void_p]
for k, a in sets.items():
    us = wall_us(a, n=1, reps=20)
    buf = (ctypes.c_ulonglong * 256)()
    assert fn(ctypes.addressof(buf)) == 0
    ends = []
    for w in range(4):
        t = list(buf[w * 64:(w + 1) * 64])
        d = [x - t[0] for x in t if 0 < x - t[0] < 10_000_000]      # (slots a wave never stamped hold an older launch's values)
        ends.append(max(d) if d else 0)
    cyc = max(ends)
    print("%-13s trace build: %.2f us per launch alone; last stamps %s cycles after entry -> >= %.2f GHz effective (cycles / wall; the "
          "launch also holds dispatch and drain)" % (k, us, ends, cyc / us / 1e3))
