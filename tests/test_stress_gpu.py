"""BASELINE.json configs[4] at its size: N = 10 000-node graph, F in {1024, 2048}, CSR at density 4e-4 and 1e-2 and the
dense bf16 adjacency, against fp64 numpy on sampled rows (the full product is 2e11 MACs: the oracle samples rows, every
row is computed by the same code path).  Tolerances: the CSR SpMM accumulates in fp32 in ascending column order
(<= 1e-5 relative of the row's magnitude); the dense path rounds both operands to bf16 first -- compared with fp64 on the
SAME rounded operands, so only fp32 accumulation order remains (<= 2e-4 of the row magnitude over K = 10 000)."""
import numpy as np
import pytest
import torch

from mgnns_amd import ops, stress

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = stress.N_NODES


def _lrelu(x):
    return np.where(x > 0, x, 0.2 * x)


@pytest.mark.parametrize("density", stress.DENSITIES)
@pytest.mark.parametrize("F", [1024, 2048])
def test_csr_spmm_10k_nodes(density, F):
    rp, col, val = stress.random_csr(N, density, seed=3)
    csr = stress.csr_to_device((rp, col, val), DEV)
    g = torch.Generator(device=DEV).manual_seed(F)
    X = torch.randn(N, F, device=DEV, generator=g)
    Y = ops.spmm_csr(csr, X, act=ops.ACT_LRELU2)
    Y2 = torch.empty_like(Y)
    assert ops.spmm_csr(csr, X, act=ops.ACT_LRELU2, out=Y2) is Y2 and torch.equal(Y, Y2)      # deterministic, out= honoured
    Xh = X.cpu().numpy().astype(np.float64)
    rows = np.unique(np.concatenate([[0, 1, N - 1], np.random.RandomState(1).randint(0, N, 60),
                                     np.argsort(np.diff(rp))[-3:], np.argsort(np.diff(rp))[:3]]))
    Yh = Y[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    for i, r in enumerate(rows):
        lo, hi = rp[r], rp[r + 1]
        ref = _lrelu((val[lo:hi].astype(np.float64)[:, None] * Xh[col[lo:hi]]).sum(0))
        scale = np.abs(val[lo:hi]).astype(np.float64) @ np.abs(Xh[col[lo:hi]]) + 1e-30
        assert np.max(np.abs(Yh[i] - ref) / scale.max()) < 1e-5, "row %d" % r
    assert torch.isfinite(Y).all()


@pytest.mark.parametrize("F", [1024, 2048])
def test_dense_bf16_adjacency_10k_nodes(F):
    g = torch.Generator(device=DEV).manual_seed(7 + F)
    adj = torch.rand(N, N, device=DEV, generator=g) * (2.0 / N)
    S = torch.randn(N, F, device=DEV, generator=g)
    kp = (N + 63) // 64 * 64
    adj_bf = ops.cast_pad_bf16(adj, ld=kp)
    assert adj_bf.shape == (N, kp) and float(adj_bf[:, N:].abs().max()) == 0.0
    Y = ops.dense_adj_matmul_bf16(adj_bf, S, act=ops.ACT_LRELU2)
    rows = np.unique(np.concatenate([[0, 255, 256, N - 1], np.random.RandomState(2).randint(0, N, 28)]))
    ridx = torch.from_numpy(rows).to(DEV)
    A_r = adj_bf[ridx][:, :N].float().cpu().numpy().astype(np.float64)            # the rounded operands, exactly
    S_r = S.bfloat16().float().cpu().numpy().astype(np.float64)
    ref = _lrelu(A_r @ S_r)
    scale = np.abs(A_r) @ np.abs(S_r)
    err = np.abs(Y[ridx].cpu().numpy() - ref) / scale.max()
    assert err.max() < 2e-4
    # and the rounding itself: within bf16 operand precision of the unrounded fp64 product
    full = _lrelu(adj[ridx].cpu().numpy().astype(np.float64) @ S.cpu().numpy().astype(np.float64))
    assert np.abs(Y[ridx].cpu().numpy() - full).max() / np.abs(full).max() < 2e-2


def _bf16_ref(rp, col, vbf, Xbf, rows, act=True):
    """fp64 product on the ROUNDED operands (what the kernels are given), sampled rows -> (ref, scale)."""
    Xh = Xbf.float().cpu().numpy().astype(np.float64)
    vh = vbf.float().cpu().numpy().astype(np.float64)
    ref, scale = [], []
    for r in rows:
        lo, hi = rp[r], rp[r + 1]
        y = (vh[lo:hi, None] * Xh[col[lo:hi]]).sum(0) if hi > lo else np.zeros(Xh.shape[1])
        ref.append(_lrelu(y) if act else y)
        scale.append((np.abs(vh[lo:hi]) @ np.abs(Xh[col[lo:hi]])).max() + 1e-30 if hi > lo else 1.0)
    return np.array(ref), np.array(scale)[:, None]


@pytest.mark.parametrize("density", stress.DENSITIES)
@pytest.mark.parametrize("F", [1024, 2048])
def test_bf16_spmm_10k_nodes(density, F):
    """configs[4] (ii) in its stated dtype: bf16 adjacency values and features, fp32 accumulation, at N = 10 000, on the path
    `ops.spmm_bf16` picks for the density (ring / register gathers at 4e-4, LDS-tiled at 1e-2), bf16 and fp32 outputs."""
    rp, col, val = stress.random_csr(N, density, seed=3)
    adj = ops.SparseAdjBf16(stress.csr_to_device((rp, col, val), DEV))
    assert (adj.avg_nnz >= ops.SparseAdjBf16.TILED_MIN_AVG_NNZ) == (density >= 1e-3)
    g = torch.Generator(device=DEV).manual_seed(F)
    X = torch.randn(N, F, device=DEV, generator=g).bfloat16()
    rows = np.unique(np.concatenate([[0, 1, N - 1], np.random.RandomState(1).randint(0, N, 60),
                                     np.argsort(np.diff(rp))[-3:], np.argsort(np.diff(rp))[:3]]))
    ref, scale = _bf16_ref(rp, col, adj.val, X, rows)
    ridx = torch.from_numpy(rows).to(DEV)
    Y32 = ops.spmm_bf16(adj, X, act=ops.ACT_LRELU2, out_dtype=torch.float32)
    assert np.max(np.abs(Y32[ridx].cpu().numpy() - ref) / scale) < 1e-6          # fp32 accumulation order only
    Y16 = ops.spmm_bf16(adj, X, act=ops.ACT_LRELU2)
    assert Y16.dtype == torch.bfloat16 and torch.equal(Y16, Y32.bfloat16())        # the bf16 output is the rounded fp32 one
    Y2 = torch.empty_like(Y16)
    assert ops.spmm_bf16(adj, X, act=ops.ACT_LRELU2, out=Y2) is Y2 and torch.equal(Y2, Y16)   # deterministic, out= honoured
    assert torch.isfinite(Y32).all()
    # the gather forms compute the same fmaf chain per row: bit-identical results; the LDS-tiled kernel accumulates with
    # v_dot2c_f32_bf16 (same order, its own rounding of the last bit)
    if density >= 1e-3:
        Yd = ops.spmm_bf16(adj, X, act=ops.ACT_LRELU2, out_dtype=torch.float32, path="direct")
        assert float((Yd - Y32).abs().max()) <= 1e-6 * float(Y32.abs().max())
    else:
        for var in ((1 << 30) | (384 << 4) | 4, (1 << 30) | (256 << 4) | 1, (1 << 29) | (64 << 8), (1 << 29) | (128 << 8) | 3):
            assert torch.equal(ops.spmm_bf16(adj, X, act=ops.ACT_LRELU2, out_dtype=torch.float32, path="direct", variant=var), Y32), var


@pytest.mark.parametrize("geo", [(8, 10, 128), (8, 20, 128), (4, 10, 256), (4, 20, 256)])
def test_bf16_spmm_tiled_geometries_ragged_and_multipart(geo):
    """Every built geometry of the LDS-tiled kernel on a small rectangular problem with empty rows, a full row (records of
    several parts), row / column counts that are not multiples of the block sizes -- against the full fp64 product."""
    rs = np.random.RandomState(geo[1])
    n, ncols, F = 205, 390, 512
    dense = (rs.rand(n, ncols) < 0.1) * rs.rand(n, ncols)
    dense[7] = 0
    dense[n - 1] = 0
    dense[11, :] = rs.rand(ncols)
    dense[12, :] = rs.rand(ncols)
    rp = np.concatenate([[0], np.cumsum((dense != 0).sum(1))]).astype(np.int32)
    col = np.nonzero(dense)[1].astype(np.int32)
    val = dense[dense != 0].astype(np.float32)
    adj = ops.SparseAdjBf16(stress.csr_to_device((rp, col, val), DEV), n_cols=ncols)
    X = torch.from_numpy(rs.randn(ncols, F).astype(np.float32)).to(DEV).bfloat16()
    ref, scale = _bf16_ref(rp, col, adj.val, X, range(n), act=False)
    for path, kw in (("tiled", dict(geometry=geo)), ("direct", {})):
        Y = ops.spmm_bf16(adj, X, out_dtype=torch.float32, path=path, **kw).cpu().numpy()
        assert np.max(np.abs(Y - ref) / scale) < 1e-6, path
        assert not Y[7].any() and not Y[n - 1].any()
    a, b = ops.spmm_bf16(adj, X, out_dtype=torch.float32, path="tiled", geometry=geo), ops.spmm_bf16(adj, X, out_dtype=torch.float32, path="direct")
    assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    assert torch.equal(a, ops.spmm_bf16(adj, X, out_dtype=torch.float32, path="tiled", geometry=geo))     # deterministic


def test_bf16_spmm_direct_edge_cases():
    """Odd feature widths (partial last slab), odd nnz (values travel as dwords), a single row, an all-empty matrix, and
    argument errors."""
    rs = np.random.RandomState(9)
    for n, F, dens in ((37, 200, 0.3), (1, 8, 1.0), (300, 136, 0.02), (64, 1096, 0.1)):
        rp, col, val = stress.random_csr(n, dens, seed=n)
        if col.size % 2 == 0:                                      # force an odd number of non-zeros
            rp, col, val = rp.copy(), col[:-1], val[:-1]
            rp[-1] -= 1
        adj = ops.SparseAdjBf16(stress.csr_to_device((rp, col, val), DEV))
        X = torch.from_numpy(rs.randn(n, F).astype(np.float32)).to(DEV).bfloat16()
        ref, scale = _bf16_ref(rp, col, adj.val, X, range(n), act=False)
        for var in (0, (1 << 30) | (8 << 4), (1 << 30) | (8 << 4) | 5, (1 << 29) | (2 << 8), (1 << 29) | (3 << 8) | 3):
            Y = ops.spmm_bf16(adj, X, out_dtype=torch.float32, path="direct", variant=var).cpu().numpy()
            assert np.max(np.abs(Y - ref) / scale) < 1e-6, (n, F, var)
    empty = ops.SparseAdjBf16((torch.zeros(6, dtype=torch.int32, device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV),
                               torch.zeros(0, device=DEV)))
    X = torch.ones(5, 16, device=DEV).bfloat16()
    for var in (0, (1 << 30) | (8 << 4)):
        assert not ops.spmm_bf16(empty, X, path="direct", variant=var).float().abs().sum().item()
    rp, col, val = stress.random_csr(16, 0.5, 1)
    adj = ops.SparseAdjBf16(stress.csr_to_device((rp, col, val), DEV))
    with pytest.raises(TypeError):
        ops.spmm_bf16(adj, torch.ones(16, 16, device=DEV))                       # fp32 features
    with pytest.raises(ValueError):
        ops.spmm_bf16(adj, torch.ones(15, 16, device=DEV).bfloat16())            # wrong row count
    with pytest.raises(RuntimeError):
        ops.spmm_bf16(adj, torch.ones(16, 12, device=DEV).bfloat16())            # F % 8
    with pytest.raises(RuntimeError):
        ops.spmm_bf16(adj, torch.ones(16, 256, device=DEV).bfloat16(), path="tiled", geometry=(8, 12, 128))   # no such kernel


def _channel_fp64(ch, pooled, A):
    X, W1, W2 = (t.cpu().numpy().astype(np.float64) for t in (ch.X, ch.W1, ch.W2))
    G = A @ (_lrelu(A @ (X @ W1)) @ W2)
    return pooled.cpu().numpy().astype(np.float64) @ G.T


def test_stress_channel_end_to_end_small_vs_fp64():
    """The whole channel (X.W1 -> adj -> LeakyReLU -> .W2 -> adj -> read-out) at a size fp64 can do in full: fp32 features
    (1e-5) and configs[4]'s bf16 chain, sparse and dense (four bf16 roundings of intermediates: <= 2e-2 of the output scale)."""
    import scipy.sparse as sp
    n, B = 1500, 32
    pooled = torch.relu(torch.randn(B, 2048, device=DEV))
    ch = stress.StressChannel(n=n, density=4e-3, seed=5, dev=DEV, dtype="f32")
    rp, col, val = (a.cpu().numpy() for a in ch.csr)
    A = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(n, n))
    ref = _channel_fp64(ch, pooled, A)
    assert np.abs(ch.forward(pooled).cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-5
    for dens in (4e-3, 5e-2):                          # the gather path and the LDS-tiled path (75 non-zeros per row)
        chb = stress.StressChannel(n=n, density=dens, seed=5, dev=DEV)
        rp, col, val = (a.cpu().numpy() for a in chb.csr)
        A = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(n, n))
        assert (chb.sadj.avg_nnz >= ops.SparseAdjBf16.TILED_MIN_AVG_NNZ) == (dens > 1e-2)
        ref = _channel_fp64(chb, pooled, A)
        out = chb.forward(pooled)
        assert out.dtype == torch.float32 and tuple(out.shape) == (B, n)
        assert np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-2, dens
    chd = stress.StressChannel(n=n, dense=True, seed=6, dev=DEV)
    ref = _channel_fp64(chd, pooled, chd.adj.cpu().numpy().astype(np.float64))
    assert np.abs(chd.forward(pooled).cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-2


def test_stress_channel_reassociated_order_small_vs_fp64():
    """order='reassociated' -- every layer as (adj . X) . W instead of the reference's adj . (X . W) (MODEL:52-58): algebraically the
    same channel, the adjacency product on the layer's input width; against the same fp64 reference under the same bf16 gate (2e-2 of
    the output scale: four bf16 roundings of intermediates either way), and within that gate of the reference order's own result."""
    import scipy.sparse as sp
    n, B = 1500, 32
    pooled = torch.relu(torch.randn(B, 2048, device=DEV))
    for dens in (4e-3, 5e-2):
        a = stress.StressChannel(n=n, density=dens, seed=5, dev=DEV)
        b = stress.StressChannel(n=n, density=dens, seed=5, dev=DEV, order="reassociated")
        rp, col, val = (t.cpu().numpy() for t in b.csr)
        A = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(n, n))
        ref = _channel_fp64(b, pooled, A)
        oa, ob = a.forward(pooled).cpu().numpy(), b.forward(pooled).cpu().numpy()
        ea, eb = np.abs(oa - ref).max() / np.abs(ref).max(), np.abs(ob - ref).max() / np.abs(ref).max()
        print("sparse d=%g: reference order %.2e, reassociated %.2e of the output scale" % (dens, ea, eb))
        assert eb < 2e-2 and np.abs(oa - ob).max() / np.abs(ref).max() < 2e-2
    a = stress.StressChannel(n=n, dense=True, seed=6, dev=DEV)
    b = stress.StressChannel(n=n, dense=True, seed=6, dev=DEV, order="reassociated")
    assert torch.equal(a.adj_bf16, b.adj_bf16)
    ref = _channel_fp64(b, pooled, b.adj.cpu().numpy().astype(np.float64))
    oa, ob = a.forward(pooled).cpu().numpy(), b.forward(pooled).cpu().numpy()
    eb = np.abs(ob - ref).max() / np.abs(ref).max()
    print("dense: reference order %.2e, reassociated %.2e of the output scale" % (np.abs(oa - ref).max() / np.abs(ref).max(), eb))
    assert eb < 2e-2
    assert torch.equal(b.forward(pooled), b.forward(pooled))                     # repeatable
    with pytest.raises(ValueError):
        stress.StressChannel(n=64, dev=DEV, dtype="f32", order="reassociated")


def test_stress_workload_shards_equal_the_single_rank_result():
    """configs[4] sharded (plan_shards): every rank's blocks, put together, are the one-rank result bit for bit."""
    n, B = 1200, 48
    one = stress.StressWorkload(0, 1, n=n, batch=B, dev=DEV).forward()
    full = {c: v for (c, b0, b1), v in one.items()}
    assert sorted(full) == [0, 1, 2] and all(tuple(v.shape) == (B, n) for v in full.values())
    for world in (2, 3, 5, 8):
        seen = {c: torch.zeros(B, dtype=torch.bool) for c in full}
        for rank in range(world):
            for (c, b0, b1), v in stress.StressWorkload(rank, world, n=n, batch=B, dev=DEV).forward().items():
                assert torch.equal(v, full[c][b0:b1]), (world, rank, c)
                assert not seen[c][b0:b1].any()
                seen[c][b0:b1] = True
        assert all(m.all() for m in seen.values())


def test_stress_workload_graphs_on_three_streams_equal_the_eager_forward():
    """StressWorkload.capture(): one hipGraph per channel replayed side by side on three streams == the eager forward, bit for
    bit, replay after replay (sparse and dense adjacency)."""
    for kw in (dict(density=stress.DENSITIES[0]), dict(dense=True)):
        wl = stress.StressWorkload(0, 1, n=1500, batch=64, dev=DEV, **kw)
        ref = {k: v.clone() for k, v in wl.forward().items()}
        replay = wl.capture()
        for _ in range(3):
            got = replay()
            torch.cuda.synchronize()
            assert set(got) == set(ref)
            for k in ref:
                assert torch.equal(got[k], ref[k]), (kw, k)


def test_stress_measure_reports_cold_figures():
    r = stress.measure(DEV, n=4000, batch=64)
    for k in ("spmm_bf16_d0.0004_F1024", "spmm_bf16_d0.01_F2048", "spmm_f32_d0.0004_F1024", "dense_adj_bf16_F1024",
              "workload_bf16_csr_d0.0004", "workload_bf16_dense", "channel_f32_csr_d0.0004"):
        assert k in r, k
    s = r["spmm_bf16_d0.0004_F2048"]
    assert s["sets"] >= 4 and s["sets"] * 2 * 4000 * 2048 * 2 >= stress.COLD_BYTES and s["cold_ms"] > 0 and 0 < s["frac_of_copy"] < 1.5
    assert s["algorithmic_MB"] == pytest.approx((s["nnz"] * 6 + 2 * 4000 * 2048 * 2) / 1e6, abs=0.01)
    assert r["spmm_bf16_d0.01_F1024"]["path"].startswith("tiled") and s["path"].startswith("direct")


def _channel_ref_fp64_device(ch, pooled_bf16):
    """fp64 on the device (torch / rocBLAS, nothing of ours) of one configs[4] channel at FULL size, twice: `exact` keeps
    fp64 between the products; `chain` rounds every intermediate to bf16 where the bf16 workload stores one (S1, X1, S2, G),
    so only the fp32 accumulation order separates it from the kernels."""
    bf, f64 = torch.bfloat16, torch.float64
    n = ch.n
    if ch.dense:
        A = ch.adj_bf16[:, :n].to(f64)
    else:
        rp, col, val = ch.csr
        rows = torch.repeat_interleave(torch.arange(n, device=rp.device), (rp[1:] - rp[:-1]).long())
        A = torch.zeros(n, n, device=rp.device, dtype=f64)
        A.index_put_((rows, col.long()), val.to(bf).to(f64), accumulate=True)      # (the bf16 SpMM holds bf16 values)
    X, W1, W2 = ch.Xb[:, :300].to(f64), ch.W1t[:, :300].to(f64).t(), ch.W2t.to(f64).t()
    P = pooled_bf16.to(f64)
    lrelu = lambda t: torch.where(t > 0, t, 0.2 * t)
    exact = P @ (A @ (lrelu(A @ (X @ W1)) @ W2)).t()
    r = lambda t: t.to(torch.float32).to(bf).to(f64)
    chain = P @ r(A @ r(r(lrelu(A @ r(X @ W1))) @ W2)).t()
    return exact, chain


@pytest.mark.parametrize("kind", ["csr_d4e-4", "csr_d1e-2", "dense", "dense_reassociated"])
def test_stress_workload_full_size(kind):
    """BASELINE.json configs[4] AS A WORKLOAD at its stated size -- N = 10 000 nodes, 3 channels, batch 512, bf16 -- the forward
    bench.py's `stress` leg times (MODEL:52-58, 460-474 at stress size): every one of the 3 x 512 x 10 000 outputs against fp64
    (not a sample), and -- for the PMI-like density -- the 3- and 8-rank shard plans against the one-rank result at that size."""
    kw = {"csr_d4e-4": dict(density=stress.DENSITIES[0]), "csr_d1e-2": dict(density=stress.DENSITIES[1]), "dense": dict(dense=True),
          "dense_reassociated": dict(dense=True, order="reassociated")}[kind]      # (the (adj . X) . W variant: same gates vs fp64)
    wl = stress.StressWorkload(0, 1, dev=DEV, **kw)
    assert [s for s in wl.shards] == [(0, 0, stress.BATCH), (1, 0, stress.BATCH), (2, 0, stress.BATCH)]
    out = wl.forward()
    torch.cuda.synchronize()
    full = {}
    for (c, b0, b1), v in out.items():
        assert v.dtype == torch.float32 and tuple(v.shape) == (stress.BATCH, N) and bool(torch.isfinite(v).all())
        exact, chain = _channel_ref_fp64_device(wl.channels[c], wl.pooled[c])
        scale = float(exact.abs().max())
        assert scale > 0
        e_chain = float((v.double() - chain).abs().max()) / scale
        e_exact = float((v.double() - exact).abs().max()) / scale
        # same roundings as the kernels' chain: what is left is fp32 accumulation order and the odd 1-ulp bf16 flip of an
        # intermediate it causes; against unrounded fp64: four bf16 roundings of intermediates
        if kind != "dense_reassociated":                 # (`chain` rounds the reference order's intermediates)
            assert e_chain < 4e-3, (kind, c, e_chain)
        assert e_exact < 2e-2, (kind, c, e_exact)
        # >= 64 sampled (sample, node) entries per channel, relative to each entry's own magnitude where it is not tiny
        g = torch.Generator(device="cpu").manual_seed(100 + c)
        bi = torch.randint(0, stress.BATCH, (256,), generator=g).to(v.device)
        ni = torch.randint(0, N, (256,), generator=g).to(v.device)
        got, want = v[bi, ni].double(), (exact if kind == "dense_reassociated" else chain)[bi, ni]
        big = want.abs() > 0.05 * scale
        assert int(big.sum()) >= 64, (kind, c, int(big.sum()))
        assert float(((got - want).abs() / want.abs())[big].max()) < 3e-2, (kind, c)
        full[c] = v
        del exact, chain
    if kind != "csr_d4e-4":
        return
    for world in (3, 8):
        seen = {c: torch.zeros(stress.BATCH, dtype=torch.bool) for c in full}
        for rank in range(world):
            for (c, b0, b1), v in stress.StressWorkload(rank, world, dev=DEV, **kw).forward().items():
                assert torch.equal(v, full[c][b0:b1]), (world, rank, c)
                assert not seen[c][b0:b1].any()
                seen[c][b0:b1] = True
        assert all(m.all() for m in seen.values())


def test_bf16_spmm_rows_in_order_of_length_and_block_diagonal_union():
    """The gather kernels take a static adjacency with its rows SORTED BY LENGTH (+ the map back to the rows of Y; the ring
    form in strided quads) and the three channels of configs[4] as ONE launch on the block-diagonal union: every row of Y is
    the same fmaf chain either way -- bit-identical to the rows-in-graph-order launch, both kernel forms (F = 1024: ring,
    F = 2048: register), bf16 and fp32 outputs, empty rows and one very long row included."""
    n = 3000
    rs = np.random.RandomState(5)
    csrs = []
    for c in range(3):
        rp, col, val = stress.random_csr(n, 2e-3, 40 + c)
        csrs.append((rp, col, val))
    # an adjacency with empty rows and one row of 500 entries
    per = rs.poisson(3.0, size=n)
    per[7] = 0
    per[8] = 500
    rows = np.repeat(np.arange(n), per)
    cols = rs.randint(0, n, size=rows.size)
    key = np.unique(rows.astype(np.int64) * n + cols)
    rows, cols = key // n, key % n
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    csrs[1] = (np.cumsum(rp).astype(np.int32), cols.astype(np.int32), rs.uniform(0, 1, cols.size).astype(np.float32))
    dev_csrs = [stress.csr_to_device(c, DEV) for c in csrs]
    plain = [ops.SparseAdjBf16(c, sort_rows=False) for c in dev_csrs]
    srt = [ops.SparseAdjBf16(c) for c in dev_csrs]
    big = ops.SparseAdjBf16.block_diagonal(srt)
    assert big.n_rows == 3 * n and big.n_cols == 3 * n and big.nnz == sum(a.nnz for a in srt)
    for F in (1024, 2048, 136):
        x = torch.randn(3 * n, F, device=DEV).bfloat16()
        for dt in (torch.bfloat16, torch.float32):
            ref = torch.cat([ops.spmm_bf16(plain[c], x[c * n:(c + 1) * n].contiguous(), act=ops.ACT_LRELU2, out_dtype=dt) for c in range(3)])
            assert srt[0].sorted_for(F) is not None
            got = torch.cat([ops.spmm_bf16(srt[c], x[c * n:(c + 1) * n].contiguous(), act=ops.ACT_LRELU2, out_dtype=dt) for c in range(3)])
            assert torch.equal(got, ref), (F, dt)
            one = ops.spmm_bf16(big, x, act=ops.ACT_LRELU2, out_dtype=dt)
            assert torch.equal(one, ref), (F, dt)
    # and the workload: the union forward == channel by channel, bit for bit
    wl = stress.StressWorkload(0, 1, n=1500, batch=32, dev=DEV)
    a, b = wl.forward(union=True), wl.forward(union=False)
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
