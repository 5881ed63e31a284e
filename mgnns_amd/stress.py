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
             "wave_tile": "64x64", "compute_waves": 8, "producer_waves": 4,
             "other_forms": {"160x256 (round 5; one round of 252 tiles for 10 000 x 1024: the F = 1024 adjacency product and the X.W products it "
                             "balances better; round 6: 64-wide K slices = whole 128-B operand lines, half the L2 requests)":
                                 {"BK": 64, "lds_stages": 3, "lds_bytes_per_stage": 53248, "wave_tile": "80x64",
                                  "compute_waves": 8, "producer_waves": 4},
                             "320x256 (round 5; one round of 256 tiles for 10 000 x 2048: the F = 2048 adjacency product, X1.W2; round 6: 64-wide "
                             "K slices, two stages)":
                                 {"BK": 64, "lds_stages": 2, "lds_bytes_per_stage": 73728, "wave_tile": "80x64", "compute_waves": 16,
                                  "producer_waves": 0},
                             "256x256 (round 4; K >= 4096 where neither of the above balances better)":
                                 {"BK": 32, "lds_stages": 4, "lds_bytes_per_stage": 32768, "wave_tile": "64x128", "compute_waves": 8,
                                  "producer_waves": 0}},
             "chosen_by": "mg_launch_gemm_bf16's estimate: rounds of the busiest XCD x (K slices x operand rows per tile-slice + a fixed "
                          "cost per tile)"}          # csrc/gemm_bf16.hip


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


def time_cold(fn, arg_sets, reps=3, graph=True):
    """Mean ms per launch of fn(*args) cycling over arg_sets (see module docstring).  graph=True (default): one rotation is
    captured into a hipGraph and `reps` replays are timed -- a 10-20 us kernel launched from Python would otherwise measure
    the host (ctypes + hipLaunchKernel ~ 7-10 us per call); best of three.  HIP events on the launch stream."""
    for a in arg_sets:
        fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not graph:
        e0.record()
        for _ in range(reps):
            for a in arg_sets:
                fn(*a)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * len(arg_sets))
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            for a in arg_sets:
                fn(*a)
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * len(arg_sets)))
    return best


def time_warm(fn, args, reps=10):
    return time_cold(fn, [args], reps, graph=False)


def _sets_for(bytes_per_set):
    return max(4, -(-COLD_BYTES // int(bytes_per_set)))


N_CHANNELS = 3


def _transposed_ok(n):
    """ops.gemm_bf16_nt(transposed_out=True) takes a product whose M is at least eight 160-row blocks (MGNNS_STRESS_TRANSPOSED=0: never)."""
    import os
    return (n + 159) // 160 >= 8 and os.environ.get("MGNNS_STRESS_TRANSPOSED", "1") != "0"

_marks = None                      # launch_times(): [(label, event)] recorded behind every launch of a channel forward


def _mark(label):
    if _marks is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _marks.append((label, ev))


class launch_times:
    """with launch_times() as t: wl.forward()  ->  t.ms = [(label, ms)]: every launch of the eager forward between HIP events on the
    current stream (the time from the previous launch's end to this one's end: its duration back to back)."""

    def __enter__(self):
        global _marks
        _marks = []
        _mark("start")
        self.ms = []
        return self

    def __exit__(self, *exc):
        global _marks
        marks, _marks = _marks, None
        torch.cuda.synchronize()
        self.ms = [(marks[i][0], round(marks[i - 1][1].elapsed_time(marks[i][1]), 4)) for i in range(1, len(marks))]
        return False


class StressChannel:
    """One label-graph channel at stress size.  forward(pooled) -> read-out [B, N] like MODEL:461-474.

    dtype "bf16" (BASELINE configs[4]): every operand of every product is bf16, accumulation fp32, and each product writes
    the K-contiguous bf16 operand of the next one (no cast / transpose pass in between):
      sparse   S1 = X.W1 -> X1 = lrelu(adj @ S1) -> S2 = X1.W2 -> G = adj @ S2 -> pooled.G^T       (GEMM, SpMM, GEMM, SpMM, GEMM)
      dense    S1^T = W1^T.X^T -> X1 = lrelu(adj . S1) -> S2^T = W2^T.X1^T -> G = adj . S2 -> pooled.G^T   (five GEMMs)
    dtype "f32": fp32 features on the exact-f32 MFMA + the fp32 CSR SpMM (the arithmetic of the model's own label GCN).

    order "reference" (default): every layer as the reference's GraphConvolution does, support = X.W first, then adj . support
    (MODEL:52-58) -- the products above.  order "reassociated" (bf16 only; a reported VARIANT, never the default): the same layer as
    (adj . X) . W -- the adjacency product runs on the layer's INPUT width (300 -> 320, 1024) instead of its output width (1024,
    2048): 3.2 x / 2 x fewer adjacency FLOPs (dense) or gathered bytes (sparse) per layer; algebraically equal, rounded differently
    (the intermediate that is stored as bf16 is adj.X instead of X.W):
      dense    AX = adj . X -> X1^T = lrelu(W1^T . AX^T) -> AX1 = adj . X1 -> G = AX1 . W2 -> pooled.G^T          (five GEMMs)
      sparse   AX = adj @ X -> X1 = lrelu(AX . W1) -> AX1 = adj @ X1 -> G = AX1 . W2 -> pooled.G^T               (SpMM, GEMM, SpMM, GEMM, GEMM)"""

    def __init__(self, n=N_NODES, density=DENSITIES[0], dense=False, seed=0, dev="cuda:0", dtype="bf16", order="reference"):
        if order not in ("reference", "reassociated") or (order == "reassociated" and dtype != "bf16"):
            raise ValueError("order must be 'reference' or (bf16 only) 'reassociated'")
        g = torch.Generator(device=dev).manual_seed(seed)
        self.n, self.dense, self.dtype, self.order = n, dense, dtype, order
        self.X = torch.randn(n, 300, device=dev, generator=g) * 0.45
        self.W1 = torch.randn(300, 1024, device=dev, generator=g) * 0.05
        self.W2 = torch.randn(1024, 2048, device=dev, generator=g) * 0.05
        if dense:
            self.adj = torch.rand(n, n, device=dev, generator=g) * (2.0 / n)
            self.kp = (n + 63) // 64 * 64
            self.adj_bf16 = ops.cast_pad_bf16(self.adj, ld=self.kp)
            if dtype == "f32":
                del self.adj
        else:
            self.csr = csr_to_device(random_csr(n, density, seed + 1), dev)
        if dtype == "bf16":
            self.Xb = ops.cast_pad_bf16(self.X, ld=320)                   # [n, 320]: K = 300 padded to five 64-wide slices
            self.W1t = ops.transpose_cast_bf16(self.W1)                   # [1024, 320]
            self.W2t = ops.transpose_cast_bf16(self.W2)                   # [2048, 1024]
            if dense:
                self.S1t = torch.zeros(1024, self.kp, device=dev, dtype=torch.bfloat16)    # K padding stays zero
                if order == "reassociated":
                    self.Xt = torch.zeros(320, self.kp, device=dev, dtype=torch.bfloat16)  # X^T, K-contiguous (rows 300.. and the K padding zero)
                    self.Xt[:300, :n] = self.Xb[:, :300].t()
                else:
                    self.S2t = torch.zeros(2048, self.kp, device=dev, dtype=torch.bfloat16)
            else:
                self.sadj = ops.SparseAdjBf16(self.csr)
                if order == "reassociated" and self.sadj.avg_nnz >= ops.SparseAdjBf16.TILED_MIN_AVG_NNZ:
                    # adj @ X runs on the layer's INPUT width, 300 -> 320 columns, and the LDS-tiled SpMM takes widths that are multiples of
                    # 256 (others fall to the gather kernel: 0.47-0.55 ms at density 1e-2 against 0.15 at F = 1024) -- pad the features and
                    # W1's K to 512
                    self.Xb = ops.cast_pad_bf16(self.X, ld=512)
                    w1 = torch.zeros(1024, 512, device=dev, dtype=torch.bfloat16)
                    w1[:, :320] = self.W1t
                    self.W1t = w1

    def gcn(self):
        bf = torch.bfloat16
        if self.dtype == "bf16" and self.order == "reassociated":
            if self.dense:
                ax = ops.gemm_bf16_nt(self.adj_bf16, self.Xt, out_dtype=bf)                      # adj . X            [n, 320]
                _mark("adj.X")
                if _transposed_ok(self.n):                                                       # X1^T = lrelu(W1^T . AX^T), as AX . W1 stored transposed
                    ops.gemm_bf16_nt(ax, self.W1t, None, ops.ACT_LRELU2, out=self.S1t[:, :self.n], transposed_out=True)
                else:
                    ops.gemm_bf16_nt(self.W1t, ax, None, ops.ACT_LRELU2, out=self.S1t[:, :self.n])
                _mark("(adj.X).W1")
                ax1 = ops.gemm_bf16_nt(self.adj_bf16, self.S1t, out_dtype=bf)                    # adj . X1           [n, 1024]
                _mark("adj.X1")
            else:
                ax = ops.spmm_bf16(self.sadj, self.Xb)
                _mark("adj@X")
                x1 = ops.gemm_bf16_nt(ax, self.W1t, None, ops.ACT_LRELU2, out_dtype=bf)
                _mark("(adj@X).W1")
                ax1 = ops.spmm_bf16(self.sadj, x1)
                _mark("adj@X1")
            G = ops.gemm_bf16_nt(ax1, self.W2t, out_dtype=bf)                                    # G [n, 2048] bf16
            _mark("(adj.X1).W2")
            return G
        if self.dtype == "bf16" and self.dense:
            # (round 6: S^T = W^T . X^T has the SHORT side as its M -- 1024 / 2048 rows of tiles that fill the chip's rounds badly -- so the
            #  product runs as X . W with the long side as M and stores its result transposed: the same K-contiguous S^T)
            tr = _transposed_ok(self.n)
            if tr:
                ops.gemm_bf16_nt(self.Xb, self.W1t, out=self.S1t[:, :self.n], transposed_out=True)
            else:
                ops.gemm_bf16_nt(self.W1t, self.Xb, out=self.S1t[:, :self.n])
            _mark("X.W1")
            x1 = ops.gemm_bf16_nt(self.adj_bf16, self.S1t, None, ops.ACT_LRELU2, out_dtype=bf)
            _mark("adj.S1")
            if tr:
                ops.gemm_bf16_nt(x1, self.W2t, out=self.S2t[:, :self.n], transposed_out=True)
            else:
                ops.gemm_bf16_nt(self.W2t, x1, out=self.S2t[:, :self.n])
            _mark("X1.W2")
            G = ops.gemm_bf16_nt(self.adj_bf16, self.S2t, out_dtype=bf)                          # G [n, 2048] bf16
            _mark("adj.S2")
            return G
        if self.dtype == "bf16":
            s1 = ops.gemm_bf16_nt(self.Xb, self.W1t, out_dtype=bf)
            _mark("X.W1")
            x1 = ops.spmm_bf16(self.sadj, s1, act=ops.ACT_LRELU2)
            _mark("adj@S1")
            s2 = ops.gemm_bf16_nt(x1, self.W2t, out_dtype=bf)
            _mark("X1.W2")
            G = ops.spmm_bf16(self.sadj, s2)
            _mark("adj@S2")
            return G
        prop = (lambda sup, act: ops.dense_adj_matmul_bf16(self.adj_bf16, sup, act=act)) if self.dense else \
               (lambda sup, act: ops.spmm_csr(self.csr, sup, act=act))
        x = prop(ops.matmul(self.X, self.W1), ops.ACT_LRELU2)
        return prop(ops.matmul(x, self.W2), ops.ACT_NONE)                     # G [n, 2048] fp32

    def forward(self, pooled):
        """pooled [B, 2048] fp32 (bf16 mode also takes it already cast) -> pooled . G^T [B, n] fp32."""
        G = self.gcn()
        if self.dtype == "bf16":
            pb = pooled if pooled.dtype == torch.bfloat16 else ops.cast_pad_bf16(pooled, ld=2048)
            out = ops.gemm_bf16_nt(pb, G)      # (measured as (G . pooled^T)^T too: 45 us against 40 -- 126 tiles and 64-B fp32 column runs: kept straight)
            _mark("read-out")
            return out
        return ops.linear(pooled, G)


def plan_shards(world, n_channels=N_CHANNELS, batch=BATCH):
    """configs[4] over `world` ranks -> per rank a list of (channel, b0, b1).  Channels are independent (MODEL:460-506 runs
    object and scene one after the other): up to n_channels ranks take whole channels; beyond that the ranks that share a
    channel split its read-out batch (each recomputes the batch-independent G -- 4 of the 5 products -- the way every rank of
    the model's forward recomputes its label GCN).  No data-path collective: the outputs are disjoint [b1 - b0, N] blocks."""
    if world < 1:
        raise ValueError("world must be >= 1")
    out = [[] for _ in range(world)]
    if world <= n_channels:
        for c in range(n_channels):
            out[c % world].append((c, 0, batch))
        return out
    groups = [[r for r in range(world) if r % n_channels == c] for c in range(n_channels)]
    for c, ranks in enumerate(groups):
        k = len(ranks)
        for i, r in enumerate(ranks):
            b0, b1 = batch * i // k, batch * (i + 1) // k
            out[r].append((c, b0, b1))
    return out


class StressWorkload:
    """The whole of configs[4] as one rank sees it: its share of the 3 channels x batch 512 (plan_shards)."""

    def __init__(self, rank=0, world=1, n=N_NODES, batch=BATCH, density=DENSITIES[0], dense=False, dev="cuda:0", dtype="bf16",
                 n_channels=N_CHANNELS, order="reference"):
        self.shards = plan_shards(world, n_channels, batch)[rank]
        g = torch.Generator(device=dev).manual_seed(1234)
        self.channels, self.pooled = {}, {}
        for c, b0, b1 in self.shards:
            if c not in self.channels:
                self.channels[c] = StressChannel(n=n, density=density, dense=dense, seed=10 * c, dev=dev, dtype=dtype, order=order)
                full = torch.relu(torch.randn(batch, 2048, device=dev, generator=torch.Generator(device=dev).manual_seed(77 + c)))
                self.pooled[c] = ops.cast_pad_bf16(full, ld=2048) if dtype == "bf16" else full
        del g

    def forward(self, union=None):
        """-> {(channel, b0, b1): read-out [b1 - b0, N] fp32}.  union (default: whenever it applies -- bf16, gather-path sparse
        adjacency, more than one channel on this rank): the channels' sparse propagations as ONE launch per layer on the
        block-diagonal union of their adjacencies (ops.SparseAdjBf16.block_diagonal); same bits as channel by channel."""
        chans = sorted(self.channels)
        ch0 = self.channels[chans[0]]
        can = (len(chans) > 1 and ch0.dtype == "bf16" and not ch0.dense and ch0.order == "reference"
               and ch0.sadj.avg_nnz < ops.SparseAdjBf16.TILED_MIN_AVG_NNZ)
        if union is None:
            union = can         # measured: 0.520 ms against 0.540 channel by channel (eager, density 4e-4)
        if not union:
            return {(c, b0, b1): self.channels[c].forward(self.pooled[c][b0:b1].contiguous()) for c, b0, b1 in self.shards}
        if not can:
            raise ValueError("the union form needs bf16 channels with a gather-path sparse adjacency, more than one of them")
        n, bf = ch0.n, torch.bfloat16
        if getattr(self, "_union", None) is None:
            dev = ch0.Xb.device
            self._union = (ops.SparseAdjBf16.block_diagonal([self.channels[c].sadj for c in chans]),
                           torch.empty(len(chans) * n, 1024, device=dev, dtype=bf), torch.empty(len(chans) * n, 2048, device=dev, dtype=bf))
        adj, S1, S2 = self._union
        for i, c in enumerate(chans):
            ops.gemm_bf16_nt(self.channels[c].Xb, self.channels[c].W1t, out=S1[i * n:(i + 1) * n])
        X1 = ops.spmm_bf16(adj, S1, act=ops.ACT_LRELU2)
        for i, c in enumerate(chans):
            ops.gemm_bf16_nt(X1[i * n:(i + 1) * n], self.channels[c].W2t, out=S2[i * n:(i + 1) * n])
        G = ops.spmm_bf16(adj, S2)
        pos = {c: i for i, c in enumerate(chans)}
        return {(c, b0, b1): ops.gemm_bf16_nt(self.pooled[c][b0:b1].contiguous(), G[pos[c] * n:(pos[c] + 1) * n])
                for c, b0, b1 in self.shards}

    def capture(self):
        """One hipGraph per shard (a channel's five launches), replayed side by side on a stream per shard: the channels are
        independent (MODEL:460-506), so nothing orders them but the join at the end.  -> replay() returning the same dict as
        forward() (static output buffers)."""
        shards = list(self.shards)
        ins = {k: self.pooled[k[0]][k[1]:k[2]].contiguous() for k in shards}
        for _ in range(2):                                   # weight packs / workspaces exist before any capture
            for k in shards:
                self.channels[k[0]].forward(ins[k])
        torch.cuda.synchronize()
        graphs, outs, streams = {}, {}, {}
        for k in shards:
            streams[k] = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[k]):
                outs[k] = self.channels[k[0]].forward(ins[k])
            graphs[k] = g
        torch.cuda.synchronize()

        def replay():
            main = torch.cuda.current_stream()
            for k in shards:
                streams[k].wait_stream(main)
                with torch.cuda.stream(streams[k]):
                    graphs[k].replay()
            for k in shards:
                main.wait_stream(streams[k])
            return outs
        self._captured = (graphs, outs, streams, ins)            # (keeps the static buffers alive)
        return replay


def measure(dev="cuda:0", n=N_NODES, batch=BATCH, quick=True):
    """Cache-cold per-kernel figures of configs[4] on one GPU -> dict (bench.py `stress`).  HBM fractions are against
    8 TB/s; `copy_*` is a plain device copy of the same algorithmic bytes through the same rotation (the practical
    streaming ceiling at this size)."""
    g = torch.Generator(device=dev).manual_seed(0)
    out = {"what": "BASELINE configs[4]: N=%d-node label graph, one of 3 identical channels, batch %d; every timed launch "
                   "rotates over >=4 operand sets totalling >%d MiB (cache-cold, hipGraph replays of one rotation)" %
                   (n, batch, COLD_BYTES >> 20), "N": n, "batch": batch, "gemm_tile": GEMM_TILE}
    # ---- (ii) sparse adjacency, cold: bf16 values + features (configs[4]'s dtype), fp32 accumulation ----
    out["spmm_bytes"] = "nnz*(4+2) + N*F*2 (X) + N*F*2 (Y)  [SURVEY 7-8]; copy_* = a plain device copy of N*F*2 bytes in and out " \
                        "through the same rotation: the ceiling ANY kernel moving these bytes has at this size"
    for dens in DENSITIES:
        csr_np = random_csr(n, dens, 1)
        csr = csr_to_device(csr_np, dev)
        adj = ops.SparseAdjBf16(csr)
        nnz = adj.nnz
        for F in (1024, 2048):
            by = nnz * 6.0 + 2.0 * n * F * 2
            k = _sets_for(2.0 * n * F * 2)
            xs = [torch.randn(n, F, device=dev, generator=g).bfloat16() for _ in range(k)]
            ys = [torch.empty_like(x) for x in xs]
            run = lambda x, y: ops.spmm_bf16(adj, x, act=ops.ACT_LRELU2, out=y)
            ms = time_cold(run, list(zip(xs, ys)))
            ms_c = time_cold(lambda d, s_: d.copy_(s_), list(zip(ys, xs)))
            out["spmm_bf16_d%g_F%d" % (dens, F)] = {
                "path": "tiled (LDS-staged X tiles)" if adj.avg_nnz >= ops.SparseAdjBf16.TILED_MIN_AVG_NNZ else "direct (L2 gathers)",
                "nnz": nnz, "sets": k, "cold_ms": round(ms, 5), "algorithmic_MB": round(by / 1e6, 2),
                "cold_GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000.0, 4),
                "copy_cold_ms": round(ms_c, 5), "copy_cold_GBps": round(2.0 * n * F * 2 / ms_c / 1e6, 1),
                "frac_of_copy": round(ms_c / ms * by / (2.0 * n * F * 2), 4)}
            del xs, ys
        # configs[4]'s THREE channels as one launch: the block-diagonal union of three such graphs on the stacked features
        if adj.avg_nnz < ops.SparseAdjBf16.TILED_MIN_AVG_NNZ:
            adjs = [adj] + [ops.SparseAdjBf16(csr_to_device(random_csr(n, dens, 1 + 10 * c), dev)) for c in (1, 2)]
            big = ops.SparseAdjBf16.block_diagonal(adjs)
            for F in (1024, 2048):
                by = big.nnz * 6.0 + 2.0 * 3 * n * F * 2
                k = _sets_for(2.0 * 3 * n * F * 2)
                xs = [torch.randn(3 * n, F, device=dev, generator=g).bfloat16() for _ in range(k)]
                ys = [torch.empty_like(x) for x in xs]
                ms = time_cold(lambda x, y: ops.spmm_bf16(big, x, act=ops.ACT_LRELU2, out=y), list(zip(xs, ys)))
                ms_c = time_cold(lambda d, s_: d.copy_(s_), list(zip(ys, xs)))
                out["spmm_bf16_3ch_d%g_F%d" % (dens, F)] = {
                    "what": "three channels' propagations as ONE launch (block-diagonal union, rows in order of length)",
                    "nnz": big.nnz, "sets": k, "cold_ms": round(ms, 5), "cold_ms_per_channel": round(ms / 3, 5),
                    "algorithmic_MB": round(by / 1e6, 2), "cold_GBps": round(by / ms / 1e6, 1),
                    "frac_of_8TBps": round(by / ms / 1e6 / 8000.0, 4), "copy_cold_ms": round(ms_c, 5),
                    "frac_of_copy": round(ms_c / ms * by / (2.0 * 3 * n * F * 2), 4)}
                del xs, ys
            del big, adjs
        # the fp32-feature kernel of the model's own label GCN (mgnns_spmm_csr_fwd) on the same graph, for reference
        F = 1024
        by = nnz * 8.0 + 2.0 * n * F * 4
        k = _sets_for(2.0 * n * F * 4)
        xs = [torch.randn(n, F, device=dev, generator=g) for _ in range(k)]
        ys = [torch.empty_like(x) for x in xs]
        ms = time_cold(lambda x, y: ops.spmm_csr(csr, x, act=ops.ACT_LRELU2, out=y), list(zip(xs, ys)))
        out["spmm_f32_d%g_F%d" % (dens, F)] = {"nnz": nnz, "sets": k, "cold_ms": round(ms, 5), "algorithmic_MB": round(by / 1e6, 1),
                                                "cold_GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000.0, 4)}
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
    # ---- the whole workload on this GPU: 3 channels x (gc1 + LeakyReLU + gc2 + read-out of 512 samples), bf16 ----
    ref_out = {}
    for name, kw in (("csr_d0.0004", dict(density=DENSITIES[0])), ("csr_d0.01", dict(density=DENSITIES[1])), ("dense", dict(dense=True)),
                     ("csr_d0.0004_reassociated", dict(density=DENSITIES[0], order="reassociated")),
                     ("csr_d0.01_reassociated", dict(density=DENSITIES[1], order="reassociated")),
                     ("dense_reassociated", dict(dense=True, order="reassociated"))):
        wl = StressWorkload(n=n, batch=batch, dev=dev, **kw)
        for _ in range(3):                                    # steady state: the caching allocator holds the forward's blocks, the clock is up
            wl.forward()                                      # (round 6: the first timed forwards of a fresh workload read ~10 % high)
        torch.cuda.synchronize()
        ms = time_warm(wl.forward, (), reps=8)
        ms_sep = time_warm(lambda: wl.forward(union=False), (), reps=8)
        ch = next(iter(wl.channels.values()))
        ms1 = time_warm(ch.gcn, (), reps=5)
        out["workload_bf16_" + name] = {"ms_per_3_channel_forward": round(ms, 4), "samples_per_s": round(batch / ms * 1e3, 1),
                                        "ms_per_3_channel_forward_channel_by_channel": round(ms_sep, 4),
                                        "ms_gcn_of_one_channel": round(ms1, 4),
                                        "what": "3 channels one after the other on one stream, eager launches, bf16 operands / fp32 "
                                                "accumulation end to end (5 launches per channel)"}
        try:                                                  # the same three channels as one hipGraph each, side by side on three streams
            replay = wl.capture()
            msg = time_warm(replay, (), reps=10)
            out["workload_bf16_" + name].update({"ms_per_3_channel_forward_graphs_on_3_streams": round(msg, 4),
                                                 "samples_per_s_graphs_on_3_streams": round(batch / msg * 1e3, 1)})
        except Exception as e:                                # capture refused: the eager figure stands
            out["workload_bf16_" + name]["graphs_on_3_streams_error"] = "%s: %s" % (type(e).__name__, e)
        # every launch of one eager 3-channel forward between HIP events (channel by channel), ms
        wl.forward(union=False)
        with launch_times() as lt:
            res = wl.forward(union=False)
        out["workload_bf16_" + name]["launches_ms"] = [[k, v] for k, v in lt.ms]
        out["workload_bf16_" + name]["launches_ms_sum"] = round(sum(v for _, v in lt.ms), 4)
        first = res[next(iter(res))]
        if name.endswith("_reassociated"):
            base = ref_out[name[:-len("_reassociated")]]
            out["workload_bf16_" + name].update({
                "what": "VARIANT, not the reference's order: every layer as (adj . X) . W instead of adj . (X . W) (MODEL:52-58) -- the adjacency "
                        "product on the layer's input width; same five launches per channel",
                "max_abs_diff_vs_reference_order_over_output_scale": round(float((first - base).abs().max() / base.abs().max()), 6),
                "parity": "tests/test_stress_gpu.py: every output against fp64 at full size, <= 2e-2 of the output scale (the reference order's gate)"})
        else:
            ref_out[name] = first.clone()
        del wl, ch, res, first
    pooled = torch.relu(torch.randn(batch, 2048, device=dev, generator=g))
    ch = StressChannel(n=n, dev=dev, density=DENSITIES[0], dtype="f32")
    out["channel_f32_csr_d0.0004"] = {"ms": round(time_warm(ch.forward, (pooled,), reps=3), 4),
                                      "what": "ONE channel with fp32 features: exact-f32 MFMA X.W + the fp32 CSR SpMM (round 2's figure)"}
    del ch
    torch.cuda.empty_cache()
    return out
