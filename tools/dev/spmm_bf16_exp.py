"""bf16 SpMM experiments (configs[4]): correctness on sampled rows, then cache-cold timings of launch-geometry variants.
Timing = hipGraph replay of one rotation over >640 MiB of operand sets (no host launch cost in the figure)."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mgnns_amd import ops, stress, spmm_plan
dev = "cuda:0"; n = 10000
what = sys.argv[1] if len(sys.argv) > 1 else "all"
g = torch.Generator(device=dev).manual_seed(0)

def lrelu(x): return np.where(x > 0, x, 0.2 * x)

def check(adj, csr_np, F, path, **kw):
    rp, col, val = csr_np
    X = (torch.randn(n, F, device=dev, generator=g)).bfloat16()
    for od in (torch.bfloat16, torch.float32):
        Y = ops.spmm_bf16(adj, X, act=ops.ACT_LRELU2, out_dtype=od, path=path, **kw)
        torch.cuda.synchronize()
        Xh = X.float().cpu().numpy().astype(np.float64)
        vh = adj.val.float().cpu().numpy().astype(np.float64)
        rows = np.unique(np.concatenate([[0, 1, n - 1], np.random.RandomState(1).randint(0, n, 40), np.argsort(np.diff(rp))[-3:], np.argsort(np.diff(rp))[:3]]))
        Yh = Y[torch.from_numpy(rows).to(dev)].float().cpu().numpy()
        worst = 0.0
        for i, r in enumerate(rows):
            lo, hi = rp[r], rp[r + 1]
            ref = lrelu((vh[lo:hi, None] * Xh[col[lo:hi]]).sum(0))
            scale = (np.abs(vh[lo:hi]) @ np.abs(Xh[col[lo:hi]])).max() + 1e-30
            worst = max(worst, float(np.max(np.abs(Yh[i] - ref)) / scale))
        print(json.dumps({"check": path, "F": F, "out": str(od), "kw": str(kw), "max_rel_err": worst, "finite": bool(torch.isfinite(Y.float()).all())}), flush=True)

def time_graph(fn, arg_sets, reps=5):
    for a in arg_sets: fn(*a)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for a in arg_sets: fn(*a)
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * len(arg_sets)))
    return best

def sets(F, dtype, k=None):
    by = 2.0 * n * F * (2 if dtype == torch.bfloat16 else 4)
    k = k or max(4, -(-stress.COLD_BYTES // int(by)))
    xs = [torch.randn(n, F, device=dev, generator=g).to(dtype) for _ in range(k)]
    ys = [torch.empty_like(x) for x in xs]
    return xs, ys

only_dense = what == "dense"
if what == "graphs":
    rs = np.random.RandomState(0)
    graphs = {"random4e-4": stress.random_csr(n, 4e-4, 1),
              "identity": (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.ones(n, np.float32)),
              "perm": (np.arange(n + 1, dtype=np.int32), rs.permutation(n).astype(np.int32), np.ones(n, np.float32)),
              "rnd4": (np.arange(0, 4 * n + 1, 4, dtype=np.int32), np.sort(rs.randint(0, n, size=(n, 4)), axis=1).reshape(-1).astype(np.int32), np.ones(4 * n, np.float32)),
              "band4": (np.arange(0, 4 * n + 1, 4, dtype=np.int32), np.sort((np.arange(n)[:, None] + np.arange(4)[None, :]) % n, axis=1).reshape(-1).astype(np.int32), np.ones(4 * n, np.float32))}
    for F in (1024, 2048):
        xs, ys = sets(F, torch.bfloat16)
        for name, csr_np in graphs.items():
            adj = ops.SparseAdjBf16(stress.csr_to_device(csr_np, dev))
            by = adj.nnz * 6.0 + 2.0 * n * F * 2
            R = 1 << 29
            variants = [("dma_w128_S32_r2", (128 << 8) | (1 << 4) | 2), ("reg_w384_ns%d" % (1 if F == 1024 else 2), (1 << 30) | (384 << 4) | (0 if F == 1024 else 4))]
            for wgx in (32, 64, 128, 256):
                for ri8 in (0, 1):
                    for nsl in ((1,) if F == 1024 else (1, 2)):
                        variants.append(("ring_w%d_ri%d_nsl%d" % (wgx, 8 if ri8 else 16, nsl), R | (wgx << 8) | (ri8 << 1) | (nsl - 1)))
            for label, var in variants:
                ms = time_graph(lambda x, y: ops.spmm_bf16(adj, x, act=ops.ACT_LRELU2, out=y, path="direct", variant=var), list(zip(xs, ys)))
                print(json.dumps({"graph": name, "F": F, "kernel": label, "us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6), "frac8": round(by / ms / 1e6 / 8000, 4)}), flush=True)
        ms = time_graph(lambda d, s_: d.copy_(s_), list(zip(ys, xs)))
        print(json.dumps({"copy": F, "us": round(ms * 1e3, 2)}), flush=True)
        del xs, ys
    sys.exit(0)
for dens in stress.DENSITIES:
    if only_dense and dens < 1e-3: continue
    csr_np = stress.random_csr(n, dens, 1)
    csr = stress.csr_to_device(csr_np, dev)
    adj = ops.SparseAdjBf16(csr)
    nnz = adj.nnz
    if what in ("all", "check"):
        check(adj, csr_np, 1024, "direct")
        check(adj, csr_np, 2048, "direct")
        check(adj, csr_np, 1024, "direct", variant=(1 << 30) | (256 << 4) | 1)
        for var in ((1 << 29) | (128 << 8) | 2, (1 << 29) | (32 << 8) | 3, (1 << 29) | (256 << 8) | 1):
            check(adj, csr_np, 2048, "direct", variant=var)
        if dens >= 1e-3:
            for geo in ((8, 10, 128), (8, 20, 128), (4, 20, 256), (4, 10, 256)):
                check(adj, csr_np, 1024, "tiled", geometry=geo)
    if what == "check": continue
    for F in (1024, 2048):
        by = nnz * 6.0 + 2.0 * n * F * 2
        xs, ys = sets(F, torch.bfloat16)
        rec = {"dens": dens, "F": F, "nnz": nnz, "alg_MB": round(by / 1e6, 2), "sets": len(xs)}
        ms = time_graph(lambda d, s_: d.copy_(s_), list(zip(ys, xs)))
        rec["copy_us"] = round(ms * 1e3, 2); rec["copy_GBps"] = round(2.0 * n * F * 2 / ms / 1e6)
        print(json.dumps(rec), flush=True)
        if dens < 1e-3:
            R = 1 << 29
            variants = [("default", 0), ("reg_w384_ns1", (1 << 30) | (384 << 4)), ("reg_w384_ns2", (1 << 30) | (384 << 4) | 4)]
            for wgx in (64, 128, 256):
                for ri8 in (0, 1):
                    for nsl in ((1,) if F == 1024 else (1, 2)):
                        variants.append(("ring_w%d_ri%d_nsl%d" % (wgx, 8 if ri8 else 16, nsl), R | (wgx << 8) | (ri8 << 1) | (nsl - 1)))
            for label, var in variants:
                if F == 1024 and label.endswith("ns2"): continue
                ms = time_graph(lambda x, y: ops.spmm_bf16(adj, x, act=ops.ACT_LRELU2, out=y, path="direct", variant=var), list(zip(xs, ys)))
                print(json.dumps({"direct": label, "dens": dens, "F": F, "us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6), "frac8": round(by / ms / 1e6 / 8000, 4)}), flush=True)
        else:
            ms = time_graph(lambda x, y: ops.spmm_bf16(adj, x, act=ops.ACT_LRELU2, out=y, path="direct"), list(zip(xs, ys)), reps=2)
            print(json.dumps({"direct": 1, "dens": dens, "F": F, "us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6), "frac8": round(by / ms / 1e6 / 8000, 4)}), flush=True)
            for geo in ((8, 10, 128), (8, 20, 128), (4, 20, 256), (4, 10, 256)):
                ms = time_graph(lambda x, y: ops.spmm_bf16(adj, x, act=ops.ACT_LRELU2, out=y, path="tiled", geometry=geo), list(zip(xs, ys)), reps=2)
                print(json.dumps({"tiled": 1, "dens": dens, "F": F, "geo": geo, "us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6), "frac8": round(by / ms / 1e6 / 8000, 4)}), flush=True)
        del xs, ys
    # the fp32 kernel of round 2 on the same graph, graph-timed, for reference
    for F in (1024,):
        xs, ys = sets(F, torch.float32)
        by = nnz * 8.0 + 2.0 * n * F * 4
        ms = time_graph(lambda x, y: ops.spmm_csr(csr, x, act=ops.ACT_LRELU2, out=y), list(zip(xs, ys)), reps=2)
        print(json.dumps({"fp32_kernel": 1, "dens": dens, "F": F, "us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6), "frac8": round(by / ms / 1e6 / 8000, 4)}), flush=True)
        del xs, ys
