"""BASELINE.json configs[4] -- "dense-adjacency stress": one label-graph GCN channel (MODEL:461-474: gen_adj'd adjacency,
GraphConvolution x2 with LeakyReLU(0.2), read-out pooled . G^T) at N = 10 000 nodes and batch 512, with the adjacency

  (i)  DENSE, kept in bf16 [N, N] (200 MB): adj @ support on the bf16 MFMA GEMM (csrc/gemm_bf16.hip), and
  (ii) CSR at PMI-like (4e-4) and dense-ish (1e-2) density: adj @ support on the slab SpMM (csrc/adj.hip).

Everything here is plumbing around the C-ABI operators (ops.*): workload construction, a channel forward, and the
cache-cold measurement used by bench.py's `stress` leg, tools/bench_stress.py and tests/test_stress_gpu.py.

Cache-cold: MI355X has a 256 MiB Infinity Cache in front of HBM; a loop over ONE operand set of 82-164 MB measures that
cache, not HBM.  `time_cold` rotates every launch over `sets` distinct (input, output) buffer sets whose total exceeds
512 MiB, so no launch finds its operands resident.  Shared, small operands (CSR metadata, 0.3 MB) may stay cached.
"""
import numpy as np
import torch

from . import ops

N_NODES = 10000
BATCH = 512
DENSITIES = (4e-4, 1e-2)
COLD_BYTES = 640 << 20            # rotate over more than 2x the 256 MiB Infinity Cache
GEMM_TILE = {"workgroup_tile": "256x128", "BK": 64, "lds_stages": 3, "lds_bytes_per_stage": 49152,
             "wave_tile": "64x64", "compute_waves": 8, "producer_waves": 4}          # csrc/gemm_bf16.hip


def random_csr(n, density, seed):
    """Random sparse [n, n] with ~density*n non-zeros per row (at least one), sorted columns, values U(0,1).
    -> (row_ptr int32 [n+1], col int32 [nnz], val float32 [nnz]) numpy."""
    rs = np.random.RandomState(seed)
    per = rs.poisson(density * n, size=n).clip(1, n)
    rows = np.repeat(np.arange(n, dtype=np.int64), per)
    cols = rs.randint(0, n, size=rows.size).astype(np.int64)
    key = np.unique(rows * n + cols)                       # duplicates within a row collapse; sorted = row-major
    rows, cols = key // n, key % n
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    val = rs.uniform(0.0, 1.0, size=cols.size).astype(np.float32)
    return np.cumsum(rp).astype(np.int32), cols.astype(np.int32), val


def csr_to_device(csr, dev):
    return tuple(torch.from_numpy(a).to(dev) for a in csr)


def time_cold(fn, arg_sets, reps=3):
    """Mean ms per launch of fn(*args) cycling over arg_sets (see module docstring); one warm-up cycle, `reps` timed
    cycles, HIP events on the launch stream."""
    for a in arg_sets:
        fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for a in arg_sets:
            fn(*a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(arg_sets))


def time_warm(fn, args, reps=10):
    return time_cold(fn, [args], reps)


def _sets_for(bytes_per_set):
    return max(4, -(-COLD_BYTES // int(bytes_per_set)))


class StressChannel:
    """One label-graph channel at stress size.  forward(pooled) -> read-out [B, N] like MODEL:461-474."""

    def __init__(self, n=N_NODES, density=DENSITIES[0], dense=False, seed=0, dev="cuda:0"):
        g = torch.Generator(device=dev).manual_seed(seed)
        self.n = n
        self.X = torch.randn(n, 300, device=dev, generator=g) * 0.45
        self.W1 = torch.randn(300, 1024, device=dev, generator=g) * 0.05
        self.W2 = torch.randn(1024, 2048, device=dev, generator=g) * 0.05
        self.dense = dense
        if dense:
            adj = torch.rand(n, n, device=dev, generator=g) * (2.0 / n)
            self.adj_bf16 = ops.cast_pad_bf16(adj, ld=(n + 63) // 64 * 64)
        else:
            self.csr = csr_to_device(random_csr(n, density, seed + 1), dev)

    def _prop(self, support, act):
        if self.dense:
            return ops.dense_adj_matmul_bf16(self.adj_bf16, support, act=act)
        return ops.spmm_csr(self.csr, support, act=act)

    def gcn(self):
        x = self._prop(ops.matmul(self.X, self.W1), ops.ACT_LRELU2)
        return self._prop(ops.matmul(x, self.W2), ops.ACT_NONE)                # G [N, 2048]

    def forward(self, pooled):
        return ops.linear(pooled, self.gcn())                                 # pooled [B,2048] . G^T -> [B, N]


def measure(dev="cuda:0", n=N_NODES, batch=BATCH, quick=True):
    """Cache-cold per-kernel figures of configs[4] on one GPU -> dict (bench.py `stress`).  HBM fractions are against
    8 TB/s; `copy_*` is a plain device copy of the same algorithmic bytes through the same rotation (the practical
    streaming ceiling at this size)."""
    g = torch.Generator(device=dev).manual_seed(0)
    out = {"what": "BASELINE configs[4]: N=%d-node label graph, one of 3 identical channels, batch %d; every timed launch "
                   "rotates over >=4 operand sets totalling >%d MiB (cache-cold); algorithmic bytes = nnz*8 + 2*N*F*4" %
                   (n, batch, COLD_BYTES >> 20), "N": n, "batch": batch, "gemm_tile": GEMM_TILE}
    # ---- (ii) CSR SpMM, cold ----
    for dens in DENSITIES:
        csr_np = random_csr(n, dens, 1)
        csr = csr_to_device(csr_np, dev)
        nnz = int(csr_np[1].size)
        for F in (1024, 2048):
            by = nnz * 8.0 + 2.0 * n * F * 4
            k = _sets_for(2.0 * n * F * 4)
            xs = [torch.randn(n, F, device=dev, generator=g) for _ in range(k)]
            ys = [torch.empty_like(x) for x in xs]                 # outputs rotate too: a recycled block would stay cached
            run = lambda x, y: ops.spmm_csr(csr, x, act=ops.ACT_LRELU2, out=y)
            ms = time_cold(run, list(zip(xs, ys)))
            ms_w = time_warm(run, (xs[0], ys[0]))
            rec = {"nnz": nnz, "sets": k, "cold_ms": round(ms, 4), "algorithmic_MB": round(by / 1e6, 1),
                   "cold_GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000.0, 4),
                   "warm_ms_same_buffers": round(ms_w, 4), "warm_GBps": round(by / ms_w / 1e6, 1)}
            if dens == DENSITIES[0]:
                ms_c = time_cold(lambda d, s: d.copy_(s), list(zip(ys, xs)))
                rec["copy_cold_GBps"] = round(2.0 * n * F * 4 / ms_c / 1e6, 1)
            out["spmm_csr_d%g_F%d" % (dens, F)] = rec
            del xs, ys
    # ---- (i) dense bf16 adjacency GEMM ----
    kp = (n + 63) // 64 * 64
    for F in (1024, 2048):
        by = n * kp * 2 + F * kp * 2 + n * F * 4
        k = 4 if not quick else 3                        # A alone is 200 MB: 3 sets = 600 MB of A
        As = [ops.cast_pad_bf16(torch.rand(n, n, device=dev, generator=g) * (2.0 / n), ld=kp) for _ in range(k)]
        Sts = [ops.transpose_cast_bf16(torch.randn(n, F, device=dev, generator=g)) for _ in range(k)]
        Cs = [torch.empty(n, F, device=dev) for _ in range(k)]
        ms = time_cold(lambda a, s, c: ops.gemm_bf16_nt(a, s, None, ops.ACT_LRELU2, out=c), list(zip(As, Sts, Cs)))
        out["dense_adj_bf16_F%d" % F] = {"sets": k, "cold_ms": round(ms, 4), "tflops": round(2.0 * n * n * F / ms / 1e9, 1),
                                         "frac_of_bf16_mfma_peak": round(2.0 * n * n * F / ms / 1e9 / 2500.0, 4),
                                         "operand_GBps": round(by / ms / 1e6, 1)}
        del As, Sts, Cs
    # ---- the channel end to end (X.W1, prop, X.W2, prop, read-out), CSR 4e-4 and dense ----
    pooled = torch.relu(torch.randn(batch, 2048, device=dev, generator=g))
    for name, kw in (("csr_d0.0004", dict(density=DENSITIES[0])), ("dense_bf16", dict(dense=True))):
        ch = StressChannel(n=n, dev=dev, **kw)
        ms = time_warm(ch.forward, (pooled,), reps=5)
        out["channel_" + name] = {"ms": round(ms, 4), "what": "gc1 + LeakyReLU + gc2 + read-out of one channel, eager launches, "
                                                              "fp32 X.W on the exact-f32 MFMA"}
        del ch
    torch.cuda.empty_cache()
    return out
