#!/usr/bin/env python
"""configs[4]-style stress of the label-graph GCN step on ONE GPU: N = 10 000-node graph, X [N,300],
W1 [300,1024], W2 [1024,2048], adjacency in CSR at PMI-like (4e-4) and dense-ish (1e-2) density, read-out
[512,2048] x [2048,N].  Reports per-kernel time and the algorithmic HBM rate of the sparse step
(nnz*8 + 2*N*F*4 bytes per SpMM), and (i) the dense-bf16 [N,N] adjacency x support GEMM.

    python tools/bench_stress.py [N] [batch]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def csr(N, density, seed):
    rs = np.random.RandomState(seed)
    per = rs.poisson(density * N, size=N).clip(1, N)
    rp = np.zeros(N + 1, np.int64)
    rp[1:] = np.cumsum(per)
    col = np.concatenate([np.sort(rs.choice(N, size=k, replace=False)) for k in per]).astype(np.int32)
    val = rs.uniform(0.0, 1.0, size=col.size).astype(np.float32)
    return (torch.from_numpy(rp.astype(np.int32)).to(DEV), torch.from_numpy(col).to(DEV), torch.from_numpy(val).to(DEV)), col.size


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    g = torch.Generator(device=DEV).manual_seed(0)
    X = torch.randn(N, 300, device=DEV, generator=g) * 0.45
    W1 = torch.randn(300, 1024, device=DEV, generator=g) * 0.05
    W2 = torch.randn(1024, 2048, device=DEV, generator=g) * 0.05
    pooled = torch.relu(torch.randn(B, 2048, device=DEV, generator=g))
    out = {"N": N, "batch": B}
    ms = timeit(lambda: ops.matmul(X, W1))
    out["xw1_ms"] = round(ms, 4); out["xw1_tflops"] = round(2.0 * N * 300 * 1024 / ms / 1e9, 1)
    S1 = ops.matmul(X, W1)
    for dens in (4e-4, 1e-2):
        c, nnz = csr(N, dens, 1)
        for F, S in ((1024, S1), (2048, None)):
            if S is None:
                S = torch.randn(N, F, device=DEV, generator=g)
            ms = timeit(lambda: ops.spmm_csr(c, S, act=ops.ACT_LRELU2))
            by = nnz * 8.0 + 2.0 * N * F * 4
            gathered = nnz * (8.0 + F * 4.0) + N * F * 4.0
            out["spmm_d%g_F%d" % (dens, F)] = {"nnz": nnz, "ms": round(ms, 4), "algorithmic_GBps": round(by / ms / 1e6, 1),
                                                 "gathered_GBps": round(gathered / ms / 1e6, 1)}
    # practical streaming ceiling at these sizes: a device-to-device copy of one [N,F] operand (read + write = the
    # same 2*N*F*4 algorithmic bytes as the SpMM, no gathers)
    for F in (1024, 2048):
        src = torch.randn(N, F, device=DEV, generator=g)
        dst = torch.empty_like(src)
        ms = timeit(lambda: dst.copy_(src))
        out["copy_F%d" % F] = {"ms": round(ms, 4), "GBps": round(2.0 * N * F * 4 / ms / 1e6, 1)}
    # (i) dense adjacency kept in bf16 ([N, Kp], 200 MB at N = 10 000): adj @ support on the bf16 MFMA GEMM
    #     (workgroup tile 256 x 128, BK = 64, three 48-KB LDS stages); the support is transposed + cast per call
    adj = torch.rand(N, N, device=DEV, generator=g) * (2.0 / N)
    adj_bf = ops.cast_pad_bf16(adj, ld=(N + 63) // 64 * 64)
    del adj
    for F in (1024, 2048):
        S = torch.randn(N, F, device=DEV, generator=g)
        ms_t = timeit(lambda: ops.transpose_cast_bf16(S))
        St = ops.transpose_cast_bf16(S)
        ms = timeit(lambda: ops.gemm_bf16_nt(adj_bf, St, None, ops.ACT_LRELU2))
        out["dense_adj_bf16_F%d" % F] = {"gemm_ms": round(ms, 4), "tflops": round(2.0 * N * N * F / ms / 1e9, 1),
                                         "transpose_cast_ms": round(ms_t, 4),
                                         "A_stream_GBps": round(adj_bf.numel() * 2 / ms / 1e6, 1)}
        del S, St
    # the dense X.W products of the layer on the same kernel (weights / activations cast per call are not timed)
    for nm, (mm, kk, nn) in (("xw1_bf16", (N, 300, 1024)), ("hw2_bf16", (N, 1024, 2048))):
        a = ops.cast_pad_bf16(torch.randn(mm, kk, device=DEV, generator=g), ld=(kk + 63) // 64 * 64)
        bt = ops.transpose_cast_bf16(torch.randn(kk, nn, device=DEV, generator=g))
        ms = timeit(lambda: ops.gemm_bf16_nt(a, bt))
        out[nm] = {"ms": round(ms, 4), "tflops": round(2.0 * mm * kk * nn / ms / 1e9, 1)}
    del adj_bf
    H1 = torch.randn(N, 1024, device=DEV, generator=g)
    ms = timeit(lambda: ops.matmul(H1, W2))
    out["hw2_ms"] = round(ms, 4); out["hw2_tflops"] = round(2.0 * N * 1024 * 2048 / ms / 1e9, 1)
    G = torch.randn(N, 2048, device=DEV, generator=g)
    ms = timeit(lambda: ops.linear(pooled, G))
    out["readout_ms"] = round(ms, 4); out["readout_tflops"] = round(2.0 * B * 2048 * N / ms / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
