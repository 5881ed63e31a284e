"""Per-operator parity: HIP kernels (through the C ABI) vs the CPU oracle and the committed
golden vectors.  fp32 tolerance: 1e-5 relative per op unless stated (summation order only)."""
import numpy as np
import pytest
import torch

from mgnns_amd import _lib, ops, synth
from oracle import golden_inputs as GI
from oracle import restatement as R
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return torch.as_tensor(x).to(DEV).contiguous()


def dparams(p):
    return {k: v.to(DEV).contiguous() for k, v in p.items()}


@pytest.mark.parametrize("M,K,N", [(1, 300, 7), (7, 300, 300), (256, 1024, 300), (80, 300, 1024), (365, 1024, 2048),
                                   (5, 365, 300), (33, 37, 65), (256, 2048, 365)])
def test_linear_and_matmul(M, K, N):
    rs = np.random.RandomState(M * 7 + K + N)
    x = rs.standard_normal((M, K)).astype(np.float32)
    w = (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rs.standard_normal(N).astype(np.float32)
    res = rs.standard_normal((M, N)).astype(np.float32)
    ref = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    y = ops.linear(dev(x), dev(w), dev(b)).cpu()
    assert H.relerr(y, ref) < 2e-6
    y = ops.linear(dev(x), dev(w), dev(b), act=ops.ACT_RELU, residual=dev(res)).cpu()
    assert H.relerr(y, torch.relu(ref) + torch.from_numpy(res).double()) < 2e-6
    y = ops.linear(dev(x), dev(w), None, act=ops.ACT_LRELU2).cpu()
    assert H.relerr(y, torch.nn.functional.leaky_relu(ref - torch.from_numpy(b).double(), 0.2)) < 2e-6
    wt = np.ascontiguousarray(w.T)                              # [K,N] layout, asymmetric on purpose
    y = ops.matmul(dev(x), dev(wt)).cpu()
    assert H.relerr(y, ref - torch.from_numpy(b).double()) < 2e-6


@pytest.mark.parametrize("B", [1, 5, 256])
def test_classifier_head_equals_cat_linear_linear(B):
    """MODEL:560-566 in eval: cat -> multi_linear_1 -> (dropout = identity) -> multi_linear_2, as one launch over the composed map."""
    rs = np.random.RandomState(B)
    feats = [rs.standard_normal((B, 300)).astype(np.float32) for _ in range(4)]
    w1 = (rs.standard_normal((600, 1200)) / np.sqrt(1200)).astype(np.float32)
    b1 = rs.standard_normal(600).astype(np.float32)
    w2 = (rs.standard_normal((3, 600)) / np.sqrt(600)).astype(np.float32)
    b2 = rs.standard_normal(3).astype(np.float32)
    cat = torch.from_numpy(np.concatenate(feats, axis=1)).double()
    ref = (cat @ torch.from_numpy(w1).double().t() + torch.from_numpy(b1).double()) @ torch.from_numpy(w2).double().t() + torch.from_numpy(b2).double()
    wc = ops.matmul(dev(w2), dev(w1))
    bc = ops.linear(dev(b1)[None, :].contiguous(), dev(w2), dev(b2))[0].contiguous()
    y = ops.classifier_head([dev(f) for f in feats], wc, bc).cpu()
    assert H.relerr(y, ref) < 2e-6
    two = ops.linear(ops.linear(dev(np.concatenate(feats, axis=1)), dev(w1), dev(b1)), dev(w2), dev(b2)).cpu()
    assert H.maxabs(y, two) < 1e-5 * max(1.0, float(two.abs().max()))
    with pytest.raises(ValueError):
        ops.classifier_head([dev(f) for f in feats[:3]], wc, bc)


def test_embedding_gather():
    rs = np.random.RandomState(1)
    table = rs.standard_normal((1000, 300)).astype(np.float32)
    idx = rs.randint(0, 1000, size=(7, 53)).astype(np.int64)
    out = ops.embedding(dev(idx), dev(table)).cpu().numpy()
    assert np.array_equal(out, table[idx])                      # a gather is bit-exact
    t2 = rs.standard_normal((50, 7)).astype(np.float32)         # D % 4 != 0 path
    i2 = rs.randint(0, 50, size=(11,)).astype(np.int64)
    assert np.array_equal(ops.embedding(dev(i2), dev(t2)).cpu().numpy(), t2[i2])


def test_gen_adj_and_csr_against_reference_goldens():
    g = H.load_golden("adjacency.npz")
    for tag in ("object", "place"):
        for t in (3, 4, 5, 6):
            key = "%s_t%02d" % (tag, t)
            A = g[key + "_A"]
            adj, (rp, col, val) = ops.gen_adj(dev(A), want_csr=True)
            adj = adj.cpu()
            assert H.relerr(adj, g[key + "_adj"]) < 1e-6
            # same sparsity pattern as the reference result
            assert np.array_equal(adj.numpy() != 0, g[key + "_adj"] != 0)
            rp, col, val = rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
            C = A.shape[0]
            dense = np.zeros((C, C), np.float32)
            for i in range(C):
                cs = col[rp[i]:rp[i + 1]]
                assert np.all(np.diff(cs) > 0)
                dense[i, cs] = val[rp[i]:rp[i + 1]]
            assert np.array_equal(dense, adj.numpy())


@pytest.mark.parametrize("N,F", [(37, 300), (500, 1000), (3000, 1024), (2500, 1348), (1500, 2048), (4000, 512)])
def test_spmm_csr_ragged_rows_both_kernels(N, F):
    """CSR SpMM vs an fp64 dense product: empty rows, rows longer than the 4-wide gather group, a feature width
    that is not a multiple of the 64-float slab / 256-float chunk.  (37,300) and (500,1000) take the wave-per-row
    kernel, the others the L2-resident slab kernel with 2, 2, 4 and 1 slabs per pass."""
    rs = np.random.RandomState(N + F)
    per = rs.poisson(4.0, size=N)
    per[::7] = 0
    per[1] = min(N, 23)
    rp = np.zeros(N + 1, np.int32)
    rp[1:] = np.cumsum(per)
    col = np.concatenate([np.sort(rs.choice(N, size=k, replace=False)) for k in per] + [np.zeros(0, np.int64)]).astype(np.int32)
    val = rs.uniform(-1.0, 1.0, size=col.size).astype(np.float32)
    x = rs.standard_normal((N, F)).astype(np.float32)
    y = ops.spmm_csr((dev(rp), dev(col), dev(val)), dev(x), act=ops.ACT_LRELU2).cpu().double()
    A = torch.zeros(N, N, dtype=torch.float64)
    for i in range(N):
        A[i, col[rp[i]:rp[i + 1]].astype(np.int64)] = torch.from_numpy(val[rp[i]:rp[i + 1]]).double()
    ref = torch.nn.functional.leaky_relu(A @ torch.from_numpy(x).double(), 0.2)
    assert y.shape == (N, F)
    assert float((y - ref).abs().max()) < 1e-5
    assert float(y[::7].abs().max()) == 0.0          # empty rows produce exact zeros


def test_graph_convolution_with_bias_both_kernels():
    """GraphConvolution(bias=True) (MODEL:40-41,55-56: `output + self.bias`, a [1,1,F] parameter the reference model never
    builds): the bias rides in the propagation's epilogue, in front of an optional activation.  Module and operator against
    an fp64 dense product, wave-per-row kernel (300 nodes) and slab kernel (10 000 nodes x 256 features)."""
    from mgnns_amd.model import GraphConvolution
    for N, Fin, Fout in ((300, 64, 1000), (10000, 32, 256)):
        rs = np.random.RandomState(N)
        per = rs.poisson(3.0, size=N)
        per[::5] = 0
        rp = np.zeros(N + 1, np.int32)
        rp[1:] = np.cumsum(per)
        col = np.concatenate([np.sort(rs.choice(N, size=k, replace=False)) for k in per] + [np.zeros(0, np.int64)]).astype(np.int32)
        val = rs.uniform(-1.0, 1.0, size=col.size).astype(np.float32)
        x = rs.standard_normal((N, Fin)).astype(np.float32)
        torch.manual_seed(N)
        gc = GraphConvolution(Fin, Fout, bias=True).to(DEV)
        assert tuple(gc.bias.shape) == (1, 1, Fout)
        csr = (dev(rp), dev(col), dev(val))
        A = torch.zeros(N, N, dtype=torch.float64)
        for i in range(N):
            A[i, col[rp[i]:rp[i + 1]].astype(np.int64)] = torch.from_numpy(val[rp[i]:rp[i + 1]]).double()
        ref = A @ (torch.from_numpy(x).double() @ gc.weight.detach().cpu().double()) + gc.bias.detach().cpu().double().reshape(1, Fout)
        with torch.no_grad():
            y = gc(dev(x), csr).cpu().double()
            ya = gc(dev(x), csr, act=ops.ACT_LRELU2).cpu().double()
        assert float((y - ref).abs().max()) < 1e-5
        assert float((ya - torch.nn.functional.leaky_relu(ref, 0.2)).abs().max()) < 1e-5
        assert torch.equal(y[::5], gc.bias.detach().cpu().double().reshape(1, Fout).expand(len(y[::5]), Fout))   # empty rows: the bias alone
        gc0 = GraphConvolution(Fin, Fout).to(DEV)
        gc0.weight.data.copy_(gc.weight.data)
        with torch.no_grad():
            assert float((gc0(dev(x), csr).cpu().double() + gc.bias.detach().cpu().double().reshape(1, Fout) - y).abs().max()) < 1e-6


def test_label_attention_with_a_mask():
    """Attention.forward(mask=...) (MODEL:118-119; never passed by the reference's own forward, so the oracle's branch is a
    restatement without a golden vector): full-shape, per-sample, per-head-dim and all-masked rows (uniform softmax)."""
    from mgnns_amd.model import Attention
    p = H.params_for({"a.w_q.weight": (300, 300), "a.w_q.bias": (300,), "a.w_k.weight": (300, 365), "a.w_k.bias": (300,),
                      "a.w_v.weight": (300, 365), "a.w_v.bias": (300,), "a.fc.weight": (300, 300), "a.fc.bias": (300,)})
    att = Attention(300, 365, 5, 0.5).eval()
    att.load_state_dict({k[2:]: v for k, v in p.items()})
    att = att.to(DEV)
    rs = np.random.RandomState(5)
    B, NLQ = 9, 3
    q = torch.from_numpy(rs.standard_normal((NLQ, 300)).astype(np.float32))
    key = torch.from_numpy(rs.standard_normal((B, 365)).astype(np.float32))
    full = torch.from_numpy((rs.uniform(size=(B, NLQ, 5, 60)) > 0.3).astype(np.int64))
    full[2] = 0                                                    # a sample with every position masked
    masks = [full, full[:, :1, :1, :].contiguous(), full[:1, :1, :1, :].contiguous(), (full[:, :, :, :1] * 0 + 1).float(),
             full.bool()]
    qd, kd = dev(q), dev(key)
    with torch.no_grad():
        plain = att(qd, kd, kd)
        for m in masks:
            want = R.label_attention(p, "a", q, key, n_heads=5, mask=m)
            got = att(qd, kd, kd, mask=dev(m))
            assert H.relerr(got.cpu(), want) < 1e-5
        assert torch.equal(att(qd, kd, kd, mask=dev(masks[3])), plain)      # an all-ones mask changes nothing
        assert H.relerr(plain.cpu(), R.label_attention(p, "a", q, key, n_heads=5)) < 1e-5


def test_image_gcn_chain_against_goldens():
    g = H.load_golden("image_gcn.npz")
    adjg = H.load_golden("adjacency.npz")
    p = dparams(H.params_for({"gc1.weight": (300, 1024), "gc2.weight": (1024, 2048)}))
    proj = torch.from_numpy(GI.gcn_projection())
    for tag, key in (("object", "object_t04"), ("place", "place_t03")):
        X, pooled = GI.image_gcn_case(tag)
        _, csr = ops.gen_adj(dev(adjg[key + "_A"]), want_csr=True)
        s1 = ops.matmul(dev(X), p["gc1.weight"])
        h1 = ops.spmm_csr(csr, s1, act=ops.ACT_LRELU2)
        s2 = ops.matmul(h1, p["gc2.weight"])
        G = ops.spmm_csr(csr, s2)
        x = ops.linear(dev(pooled), G)                           # pooled @ G^T
        assert H.relerr(G.cpu() @ proj, g[tag + "_Gproj"]) < 1e-5
        assert H.relerr(x.cpu(), g[tag + "_x"]) < 1e-5


def test_persistent_label_gcn_against_goldens_and_the_separate_operators():
    """csrc/label_gcn.hip: gen_adj + GraphConvolution x 2 (+ w_q(label query), + the split-bf16 image of G) as ONE launch.
    Exact mode: the reference's golden G / read-out (1e-5) and the separate operators to fp32 rounding (their GEMM walks K in
    a different order); split-bf16 mode: fp32-class (2e-5 of the output scale).  Repeated launches, grids of 1..64 workgroups (256 is clamped to a quarter of the CUs), a dense adjacency (every ELL slot
    full) and a single-class graph."""
    g = H.load_golden("image_gcn.npz")
    adjg = H.load_golden("adjacency.npz")
    p = dparams(H.params_for({"gc1.weight": (300, 1024), "gc2.weight": (1024, 2048)}))
    proj = torch.from_numpy(GI.gcn_projection())
    rs = np.random.RandomState(3)
    lq, wq, bq = dev(rs.standard_normal((7, 300)).astype(np.float32)), dev((0.05 * rs.standard_normal((300, 300))).astype(np.float32)), dev(rs.standard_normal(300).astype(np.float32))
    exact = ops.label_gcn_pack(p["gc1.weight"], p["gc2.weight"], split=False)
    split = ops.label_gcn_pack(p["gc1.weight"], p["gc2.weight"], split=True)
    for tag, key in (("object", "object_t04"), ("place", "place_t03")):
        X, pooled = GI.image_gcn_case(tag)
        A = dev(adjg[key + "_A"])
        _, csr = ops.gen_adj(A, want_csr=True)
        G_sep = ops.spmm_csr(csr, ops.matmul(ops.spmm_csr(csr, ops.matmul(dev(X), p["gc1.weight"]), act=ops.ACT_LRELU2), p["gc2.weight"]))
        for grid in (0, 1, 7, 256):
            G, Gp, Q = ops.label_gcn(A, dev(X), exact, want_packed_g=True, query=(lq, wq, bq), grid=grid)
            assert H.maxabs(G.cpu(), G_sep.cpu()) < 2e-6 * float(G_sep.abs().max()), (tag, grid)
            if grid:
                assert torch.equal(G, G0), (tag, grid)              # the grid size does not change a single bit
            G0 = G
            assert H.relerr(G.cpu() @ proj, g[tag + "_Gproj"]) < 1e-5
            assert H.relerr(ops.linear(dev(pooled), G).cpu(), g[tag + "_x"]) < 1e-5
            ref_p = ops.pack_weight_bf16_split(G)
            assert torch.equal(Gp[0], ref_p[0]) and torch.equal(Gp[1], ref_p[1])
            assert H.maxabs(Q.cpu(), ops.linear(lq, wq, bq).cpu()) < 1e-5
            assert all(int(ws[:256].view(torch.int32).abs().sum()) == 0 for ws in exact["_scratch"].values())   # queue counters re-armed
        for _ in range(3):
            G2, Gp2, _ = ops.label_gcn(A, dev(X), split, want_packed_g=True)
            assert H.maxabs(G2.cpu(), G_sep.cpu()) < 2e-5 * float(G_sep.abs().max()), tag
            assert torch.equal(Gp2[0], ops.pack_weight_bf16_split(G2)[0])
    # dense adjacency (all C entries of every ELL row) and C = 1
    for C in (37, 1):
        A = dev((rs.rand(C, C) + 0.1).astype(np.float32))
        X = dev(rs.standard_normal((C, 300)).astype(np.float32))
        _, csr = ops.gen_adj(A, want_csr=True)
        G_sep = ops.spmm_csr(csr, ops.matmul(ops.spmm_csr(csr, ops.matmul(X, p["gc1.weight"]), act=ops.ACT_LRELU2), p["gc2.weight"]))
        G, _, _ = ops.label_gcn(A, X, exact)
        assert H.maxabs(G.cpu(), G_sep.cpu()) < 2e-6 * float(G_sep.abs().max()), C
    with pytest.raises(ValueError):
        ops.label_gcn(dev(np.eye(5, dtype=np.float32)), dev(np.zeros((4, 300), np.float32)), exact)


def test_label_attention_against_goldens():
    g = H.load_golden("label_attention.npz")
    lq = dev(g["label_query"])
    for tag, C in (("object", 80), ("place", 365)):
        p = dparams(H.params_for(H.label_attention_shapes(tag, C)))
        key = dev(GI.label_attention_key(tag))
        a = tag + "_attention."
        Q = ops.linear(lq, p[a + "w_q.weight"], p[a + "w_q.bias"])
        K = ops.linear(key, p[a + "w_k.weight"], p[a + "w_k.bias"])
        V = ops.linear(key, p[a + "w_v.weight"], p[a + "w_v.bias"])
        x = ops.label_attn_core(Q, K, V, 5)
        y = ops.linear(x, p[a + "fc.weight"], p[a + "fc.bias"])
        assert H.maxabs(y.cpu(), g[tag + "_y"]) < 1e-5
        z = ops.linear(y, p[tag + "_linear_5.weight"], p[tag + "_linear_5.bias"]).reshape(y.shape[0], -1)
        z = ops.linear(z, p[tag + "_x_linear.weight"], p[tag + "_x_linear.bias"])
        assert H.maxabs(z.cpu(), g[tag + "_z"]) < 1e-5


def test_fused_label_tail_against_goldens_and_ragged_batches():
    """csrc/label_tail.hip (read-out + K/V projection + element-wise attention + fc.linear_5 composed + x_linear + the next
    stack's query projection in one launch) vs the reference's golden `z` (Attention -> linear_5 -> x_linear on the
    unmodified reference classes), then batches that are not a multiple of the 16-sample tile vs the oracle, with the
    read-out given (x) and computed in the kernel from the two max-pool halves and the packed label-GCN matrix."""
    g = H.load_golden("label_attention.npz")
    lq = dev(g["label_query"])
    for tag, C in (("object", 80), ("place", 365)):
        pc = H.params_for(H.label_attention_shapes(tag, C))
        p = dparams(pc)
        a = tag + "_attention."
        Q = ops.linear(lq, p[a + "w_q.weight"], p[a + "w_q.bias"])
        wc = ops.matmul(p[tag + "_linear_5.weight"], p[a + "fc.weight"])
        bc = ops.linear(p[a + "fc.bias"][None, :].contiguous(), p[tag + "_linear_5.weight"], p[tag + "_linear_5.bias"])[0].contiguous()
        packed = {"wk": ops.pack_weight_f32(p[a + "w_k.weight"]), "bk": p[a + "w_k.bias"],
                  "wv": ops.pack_weight_f32(p[a + "w_v.weight"]), "bv": p[a + "w_v.bias"],
                  "wc": ops.pack_weight_f32(wc), "bc": bc, "n5": 100, "C": C,
                  "xl": ops.pack_weight_f32(p[tag + "_x_linear.weight"]), "bxl": p[tag + "_x_linear.bias"], "n_out": 300}
        key = dev(GI.label_attention_key(tag))
        z = ops.label_tail(key, Q, 5, packed)
        assert H.maxabs(z.cpu(), g[tag + "_z"]) < 1e-5
        rs = np.random.RandomState(C)
        G = (0.05 * rs.standard_normal((C, 2048))).astype(np.float32)
        wq = (0.05 * rs.standard_normal((512, 300))).astype(np.float32)
        bq = (0.05 * rs.standard_normal(512)).astype(np.float32)
        nq = (ops.pack_weight_f32(dev(wq)), dev(bq), 512)
        Gp = ops.pack_weight_f32(dev(G))
        for B in (1, 15, 17, 50):
            x = torch.from_numpy(rs.standard_normal((B, C)).astype(np.float32))
            ref = R.label_attention_tail(pc, tag, R.label_attention(pc, tag + "_attention", torch.from_numpy(g["label_query"]), x))
            z = ops.label_tail(dev(x.numpy()), Q, 5, packed)
            assert tuple(z.shape) == (B, 300) and H.maxabs(z.cpu(), ref) < 2e-5, (tag, B)
            # read-out inside: pooled = max of two halves, x = pooled . G^T
            halves = np.maximum(rs.standard_normal((B, 2, 2048)), 0).astype(np.float32)
            pooled = halves.max(axis=1)
            xr = torch.from_numpy(pooled) @ torch.from_numpy(G).t()
            ref = R.label_attention_tail(pc, tag, R.label_attention(pc, tag + "_attention", torch.from_numpy(g["label_query"]), xr))
            z2, qh = ops.label_tail(None, Q, 5, packed, pooled=dev(halves), g_wp=Gp, next_q=nq)
            scale = float(ref.abs().max())
            assert H.maxabs(z2.cpu(), ref) < 2e-5 * max(1.0, scale), (tag, B)
            ref_q = ref @ torch.from_numpy(wq).t() + torch.from_numpy(bq)
            assert H.maxabs(qh.cpu(), ref_q) < 5e-5 * max(1.0, float(ref_q.abs().max())), (tag, B)
            z3 = ops.label_tail(None, Q, 5, packed, pooled=dev(pooled[:, None, :].copy()), g_wp=Gp)     # one part
            assert torch.equal(z3, z2)
        assert tuple(ops.label_tail(key[:0].contiguous(), Q, 5, packed).shape) == (0, 300)


def test_fused_label_tail_bf16_vs_oracle():
    """bf16-mode fused channel tail (read-out + attention + maps + next query on the bf16 MFMA) vs the fp32 oracle.
    terms = 1 (plain bf16 operands, ~3 significant digits): gate 2e-2 of the output scale (measured ~3e-3);
    terms = 3 (split-bf16, fp32-class): gate 1e-4 of the output scale."""
    g = H.load_golden("label_attention.npz")
    lq = dev(g["label_query"])
    for tag, C in (("object", 80), ("place", 365)):
        pc = H.params_for(H.label_attention_shapes(tag, C))
        p = dparams(pc)
        a = tag + "_attention."
        Q = ops.linear(lq, p[a + "w_q.weight"], p[a + "w_q.bias"])
        wc = ops.matmul(p[tag + "_linear_5.weight"], p[a + "fc.weight"])
        bc = ops.linear(p[a + "fc.bias"][None, :].contiguous(), p[tag + "_linear_5.weight"], p[tag + "_linear_5.bias"])[0].contiguous()
        sp = lambda w: ops.pack_weight_bf16_split(w.contiguous())
        packed = {"wk": sp(p[a + "w_k.weight"]), "bk": p[a + "w_k.bias"], "wv": sp(p[a + "w_v.weight"]), "bv": p[a + "w_v.bias"],
                  "wc": sp(wc), "bc": bc, "n5": 100, "C": C, "xl": sp(p[tag + "_x_linear.weight"]), "bxl": p[tag + "_x_linear.bias"],
                  "n_out": 300}
        rs = np.random.RandomState(C + 1)
        G = (0.05 * rs.standard_normal((C, 2048))).astype(np.float32)
        wq = (0.05 * rs.standard_normal((1024, 300))).astype(np.float32)
        bq = (0.05 * rs.standard_normal(1024)).astype(np.float32)
        nq = (sp(dev(wq)), dev(bq), 1024)
        Gp = sp(dev(G))
        for B in (1, 17, 64):
            halves = np.maximum(rs.standard_normal((B, 2, 2048)), 0).astype(np.float32)
            xr = torch.from_numpy(halves.max(axis=1)) @ torch.from_numpy(G).t()
            ref = R.label_attention_tail(pc, tag, R.label_attention(pc, tag + "_attention", torch.from_numpy(g["label_query"]), xr))
            ref_q = ref @ torch.from_numpy(wq).t() + torch.from_numpy(bq)
            for terms, gate in ((1, 2e-2), (3, 1e-4)):
                z, qh = ops.label_tail_bf16(dev(halves), Gp, Q, 5, packed, next_q=nq, terms=terms)
                ez = H.maxabs(z.cpu(), ref) / float(ref.abs().max())
                eq = H.maxabs(qh.cpu(), ref_q) / float(ref_q.abs().max())
                print("label_tail_bf16 %s B=%d terms=%d: rel err out %.2e, qh %.2e" % (tag, B, terms, ez, eq))
                assert ez < gate and eq < gate, (tag, B, terms, ez, eq)
                z1 = ops.label_tail_bf16(dev(halves.max(axis=1)[:, None, :].copy()), Gp, Q, 5, packed, terms=terms)
                assert torch.equal(z1, z)
                if terms == 3:     # one workgroup per tile (no cluster): same result up to the order of the four K-part sums
                    z0, q0 = ops.label_tail_bf16(dev(halves), Gp, Q, 5, packed, next_q=nq, terms=3, cluster=False)
                    assert H.maxabs(z0.cpu(), z.cpu()) < 1e-5 * float(ref.abs().max())
                    assert H.maxabs(q0.cpu(), qh.cpu()) < 1e-5 * float(ref_q.abs().max())
        # the cluster exchange: deterministic over repeated launches at a chip-filling size, ragged last tile, counters re-armed
        for B in (256, 300):
            halves = dev(np.maximum(rs.standard_normal((B, 2, 2048)), 0).astype(np.float32))
            z, qh = ops.label_tail_bf16(halves, Gp, Q, 5, packed, next_q=nq, terms=3)
            for _ in range(5):
                z2, qh2 = ops.label_tail_bf16(halves, Gp, Q, 5, packed, next_q=nq, terms=3)
                assert torch.equal(z2, z) and torch.equal(qh2, qh)
            assert all(int(ws[2].abs().sum()) == 0 for ws in packed["_cluster_ws"].values())
            z0, q0 = ops.label_tail_bf16(halves, Gp, Q, 5, packed, next_q=nq, terms=3, cluster=False)
            assert H.maxabs(z0.cpu(), z.cpu()) < 1e-5 * float(z0.abs().max())
            assert H.maxabs(q0.cpu(), qh.cpu()) < 1e-5 * float(q0.abs().max())


def test_layernorm_against_golden():
    g = H.load_golden("layernorm.npz")
    p = dparams(H.params_for({"ln.gamma": (300,), "ln.beta": (300,)}))
    y = ops.layernorm(dev(g["x"]), p["ln.gamma"], p["ln.beta"]).cpu()
    rows = [r for r in range(g["x"].shape[0]) if r != 3]
    assert H.maxabs(y[rows], g["y"][rows]) < 2e-6
    # row 3 is constant: (x-mean)/(std+eps) is 0/0-like there and amplifies the rounding of the mean by
    # 1/eps = 1e6 in the reference itself, so only boundedness is comparable (|y-beta| <= |gamma|*sqrt(D))
    assert torch.isfinite(y[3]).all()
    assert float((y[3] - p["ln.beta"].cpu()).abs().max()) <= float(p["ln.gamma"].abs().max()) * 300 ** 0.5


@pytest.mark.parametrize("Hn,tag,L,masked", GI.MHA_CASES)
def test_sq_mha_core_against_goldens(Hn, tag, L, masked):
    g = H.load_golden("mha.npz")
    name = "h%d_%s" % (Hn, tag)
    pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
    p = dparams(pc)
    q, bank, mask = GI.mha_case(Hn, tag, L, masked)
    a = name + ".slf_attn."
    qh = ops.linear(dev(q), p[a + "w_qs.weight"], p[a + "w_qs.bias"])
    o, attn = ops.sq_mha_core(qh, dev(bank), None if mask is None else dev(mask), Hn, 128,
                              p[a + "w_ks.weight"], p[a + "w_ks.bias"], p[a + "w_vs.weight"], p[a + "w_vs.bias"])
    assert H.maxabs(attn.cpu(), g[name + "_attn"]) < 1e-5
    # finish the layer with the remaining ops and compare with the reference layer output
    y = ops.linear(o, p[a + "fc.weight"], p[a + "fc.bias"], residual=dev(q))
    y = ops.layernorm(y, p[a + "layer_norm.gamma"], p[a + "layer_norm.beta"])
    f = name + ".pos_ffn."
    h1 = ops.linear(y, p[f + "w_1.weight"].squeeze(-1).contiguous(), p[f + "w_1.bias"], act=ops.ACT_RELU)
    z = ops.linear(h1, p[f + "w_2.weight"].squeeze(-1).contiguous(), p[f + "w_2.bias"], residual=y)
    out = ops.layernorm(z, p[f + "layer_norm.gamma"], p[f + "layer_norm.beta"])
    assert H.maxabs(out.cpu(), g[name + "_out"]) < 2e-5


@pytest.mark.parametrize("Hn,tag,L,masked", GI.MHA_CASES)
def test_sq_mha_folded_against_goldens(Hn, tag, L, masked):
    """Folded attention (K/V projections folded into the query) reproduces the reference's attention weights and
    layer output (same goldens, same bounds as the faithful fp32 kernel) and agrees with the faithful kernel's o."""
    g = H.load_golden("mha.npz")
    name = "h%d_%s" % (Hn, tag)
    pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
    p = dparams(pc)
    q, bank, mask = GI.mha_case(Hn, tag, L, masked)
    dm = None if mask is None else dev(mask)
    a = name + ".slf_attn."
    qh = ops.linear(dev(q), p[a + "w_qs.weight"], p[a + "w_qs.bias"])
    o, attn = ops.sq_mha_folded(qh, dev(bank), dm, Hn, 128, p[a + "w_ks.weight"], p[a + "w_vs.weight"],
                                p[a + "w_vs.bias"])
    assert H.maxabs(attn.cpu(), g[name + "_attn"]) < 1e-5
    o_f, _ = ops.sq_mha_core(qh, dev(bank), dm, Hn, 128, p[a + "w_ks.weight"], p[a + "w_ks.bias"],
                             p[a + "w_vs.weight"], p[a + "w_vs.bias"])
    assert H.relerr(o.cpu(), o_f.cpu()) < 1e-5
    y = ops.linear(o, p[a + "fc.weight"], p[a + "fc.bias"], residual=dev(q))
    y = ops.layernorm(y, p[a + "layer_norm.gamma"], p[a + "layer_norm.beta"])
    f = name + ".pos_ffn."
    h1 = ops.linear(y, p[f + "w_1.weight"].squeeze(-1).contiguous(), p[f + "w_1.bias"], act=ops.ACT_RELU)
    z = ops.linear(h1, p[f + "w_2.weight"].squeeze(-1).contiguous(), p[f + "w_2.bias"], residual=y)
    out = ops.layernorm(z, p[f + "layer_norm.gamma"], p[f + "layer_norm.beta"])
    assert H.maxabs(out.cpu(), g[name + "_out"]) < 2e-5
    # without the attn output, and from the bf16 copy of the bank (exact on the bf16-rounded bank)
    o2, none = ops.sq_mha_folded(qh, dev(bank), dm, Hn, 128, p[a + "w_ks.weight"], p[a + "w_vs.weight"],
                                 p[a + "w_vs.bias"], want_attn=False)
    assert none is None and torch.equal(o2, o)
    bank_bf = ops.cast_pad_bf16(dev(bank))
    o3, attn3 = ops.sq_mha_folded(qh, bank_bf, dm, Hn, 128, p[a + "w_ks.weight"], p[a + "w_vs.weight"], p[a + "w_vs.bias"])
    o4, attn4 = ops.sq_mha_folded(qh, bank_bf[..., :300].float().contiguous(), dm, Hn, 128, p[a + "w_ks.weight"],
                                  p[a + "w_vs.weight"], p[a + "w_vs.bias"])
    assert H.maxabs(attn3.cpu(), attn4.cpu()) < 1e-6 and H.relerr(o3.cpu(), o4.cpu()) < 1e-6


def test_sq_mha_folded_ragged_lengths_and_fully_masked_tail():
    """L not a multiple of 16, L < 8 tiles (idle waves), masks that blank whole 16-row tiles."""
    rs = np.random.RandomState(77)
    for L, Hn, B in ((1, 8, 3), (17, 4, 2), (100, 8, 5), (208, 1, 2)):
        qh = dev(rs.standard_normal((B, Hn * 128)).astype(np.float32))
        bank = dev(rs.standard_normal((B, L, 300)).astype(np.float32))
        wk = dev((0.05 * rs.standard_normal((Hn * 128, 300))).astype(np.float32))
        wv = dev((0.05 * rs.standard_normal((Hn * 128, 300))).astype(np.float32))
        bk = dev(rs.standard_normal(Hn * 128).astype(np.float32))
        bv = dev(rs.standard_normal(Hn * 128).astype(np.float32))
        mask = np.ones((B, L), np.float32)
        for b in range(B):
            mask[b, rs.randint(1, L + 1):] = 0.0
        mask[0, :] = 1.0
        dm = dev(mask)
        o, attn = ops.sq_mha_folded(qh, bank, dm, Hn, 128, wk, wv, bv)
        o_f, attn_f = ops.sq_mha_core(qh, bank, dm, Hn, 128, wk, bk, wv, bv)
        assert H.maxabs(attn.cpu(), attn_f.cpu()) < 1e-5, L
        assert H.relerr(o.cpu(), o_f.cpu()) < 1e-5, L


@pytest.mark.parametrize("ngram", [1, 4])
def test_textgcn_against_goldens(ngram):
    g = H.load_golden("text_gcn.npz")
    V, count = int(g["V"]), int(g["count"])
    pmi, _ = synth.synth_pmi(V, per_row=8, seed=int(g["pmi_seed"]))
    p = dparams(H.params_for({"text_features.node_hidden.weight": (V, 300),
                              "text_features.seq_edge_w.weight": (count, 1)}))
    y = ops.textgcn(dev(g["ng%d_tok" % ngram]), p["text_features.node_hidden.weight"],
                    p["text_features.seq_edge_w.weight"], pmi.device_arrays(DEV), ngram)
    assert H.relerr(y.cpu(), g["ng%d_out" % ngram]) < 1e-5


def test_textgcn_vs_oracle_full_length_and_edge_cases():
    V, T, B = 5000, 100, 33
    pmi, count = synth.synth_pmi(V, seed=5)
    tok, lens, _ = synth.synth_tokens(B, T, V, pmi, seed=11)
    tok[3] = 0
    tok[3, 50] = 9                                              # lone token after leading PADs
    tok[4, ::2] = 0                                             # interleaved PADs
    tok[5, :] = tok[5, 0]                                       # one distinct token repeated
    p = H.params_for({"text_features.node_hidden.weight": (V, 300), "text_features.seq_edge_w.weight": (count, 1)})
    for ngram in (0, 2, 5):
        ref = R.text_gcn(tok, p["text_features.node_hidden.weight"], p["text_features.seq_edge_w.weight"], pmi, ngram)
        y = ops.textgcn(dev(tok), dev(p["text_features.node_hidden.weight"]),
                        dev(p["text_features.seq_edge_w.weight"]), pmi.device_arrays(DEV), ngram)
        assert H.relerr(y.cpu(), ref) < 1e-5


def test_textgcn_launch_forms_short_and_long_documents():
    """From 64 documents on the text GCN runs as ONE launch of the lean kernel (round 5: 256 threads, node rows read from L2 instead
    of staged in LDS, so that it fits beside an image-bank workgroup; ngram 4 = its unrolled window, 0 / 6 = its run-time window);
    rounds 3-4 ran two launches (documents of at most 24 tokens four to a CU, then the longer ones on the 1024-thread form), small
    batches run one 1024-thread launch.  Every form (ops.textgcn_set_form) on lengths on both sides of the cap incl. exactly
    24 / 25, empty and full-length documents, PADs inside: against the oracle, against each other (same values up to the summation
    order of the chunk sums) and against the same documents in batches below the threshold; repeated launches give identical bits."""
    V, T, B = 5000, 100, 96
    pmi, count = synth.synth_pmi(V, seed=6)
    tok, lens, _ = synth.synth_tokens(B, T, V, pmi, seed=12)
    rs = np.random.RandomState(3)
    for b, n in ((2, 24), (3, 25), (4, 0), (5, 100), (6, 1), (7, 23), (8, 26)):
        tok[b] = 0
        tok[b, :n] = rs.randint(2, V, size=n)
    tok[9, ::3] = 0                                             # PADs inside: the cap counts non-PAD tokens
    p = H.params_for({"text_features.node_hidden.weight": (V, 300), "text_features.seq_edge_w.weight": (count, 1)})
    nh, ew = dev(p["text_features.node_hidden.weight"]), dev(p["text_features.seq_edge_w.weight"])
    try:
        for ngram in (4, 0, 6):
            ref = R.text_gcn(tok, p["text_features.node_hidden.weight"], p["text_features.seq_edge_w.weight"], pmi, ngram)
            ys = {}
            for form in (0, 1, 2, 3):
                ops.textgcn_set_form(form)
                y = ys[form] = ops.textgcn(dev(tok), nh, ew, pmi.device_arrays(DEV), ngram)
                assert H.relerr(y.cpu(), ref) < 1e-5, (ngram, form)
                assert float(y[4].abs().max()) == 0.0
                assert torch.equal(y, ops.textgcn(dev(tok), nh, ew, pmi.device_arrays(DEV), ngram))
            assert torch.equal(ys[0], ys[3])                     # B >= 64: the default IS the lean launch
            for form in (1, 2):
                assert H.relerr(ys[form].cpu(), ys[3].cpu()) < 1e-6, (ngram, form)
            ops.textgcn_set_form(0)
            parts = torch.cat([ops.textgcn(dev(tok[i:i + 32]), nh, ew, pmi.device_arrays(DEV), ngram) for i in range(0, B, 32)])
            assert H.relerr(ys[0].cpu(), parts.cpu()) < 1e-6, ngram
            assert torch.equal(parts, ys[1])                     # B < 64: the default is the one 1024-thread launch
    finally:
        ops.textgcn_set_form(0)


def test_textgcn_explicit_ids_long_rows_and_odd_width():
    """(a) a PMI map whose ids are NOT positional (shuffled ids: the kernel must read `eid`), (b) rows far longer than the
    lookup's 9-candidate window (bisection first), hit and miss, first / last column, (c) hidden width not a multiple of 4
    (scalar gather, padded LDS rows), (d) B = 0."""
    from mgnns_amd.pmi import PmiCsr
    V, T, B = 400, 64, 9
    rs = np.random.RandomState(21)
    rows, cols = [], []
    for u in range(2, V):
        k = 120 if u % 7 == 0 else rs.randint(0, 12)              # long rows for every 7th word
        c = np.sort(rs.choice(np.arange(2, V), size=k, replace=False))
        rows += [u] * k
        cols += list(c)
    nnz = len(rows)
    eids = rs.permutation(np.arange(1, nnz + 1))                  # explicit, non-positional ids
    pmi = PmiCsr.from_coo(rows, cols, eids, V)
    assert not pmi.eid_is_positional() and pmi.device_arrays(DEV)[2] is not None
    pos = PmiCsr(pmi.row_ptr, pmi.col, np.arange(1, nnz + 1), V)
    assert pos.eid_is_positional() and pos.device_arrays(DEV)[2] is None
    count = nnz + 1
    tok = rs.randint(2, V, size=(B, T)).astype(np.int64)
    tok[0, :] = 7 * (1 + rs.randint(0, 50, size=T))               # sources with long rows
    tok[1, 10:] = 0
    tok[2, :] = 0                                                  # empty document -> zeros
    for D in (300, 150, 7):
        p = H.params_for({"text_features.node_hidden.weight": (V, D), "text_features.seq_edge_w.weight": (count, 1)})
        for m in (pmi, pos):
            for ngram in (4, 6):
                ref = R.text_gcn(tok, p["text_features.node_hidden.weight"], p["text_features.seq_edge_w.weight"], m, ngram)
                y = ops.textgcn(dev(tok), dev(p["text_features.node_hidden.weight"]),
                                dev(p["text_features.seq_edge_w.weight"]), m.device_arrays(DEV), ngram)
                assert H.relerr(y.cpu(), ref) < 1e-5, (D, ngram)
                assert float(y[2].abs().max()) == 0.0
    y0 = ops.textgcn(dev(tok[:0]), dev(p["text_features.node_hidden.weight"]), dev(p["text_features.seq_edge_w.weight"]),
                     pos.device_arrays(DEV), 4)
    assert tuple(y0.shape) == (0, 7)


@pytest.mark.parametrize("B", [1, 3])
def test_imgbank_pool_vs_oracle(B):
    rs = np.random.RandomState(3 + B)
    feat = np.maximum(rs.standard_normal((B, 2048, 196)), 0).astype(np.float32)
    w = (0.05 * rs.standard_normal((300, 2048))).astype(np.float32)
    bias = (0.05 * rs.standard_normal(300)).astype(np.float32)
    wt = ops.transpose_pad(dev(w), ops.IMGBANK_LDW)
    assert np.array_equal(wt.cpu().numpy()[:, :300], w.T) and float(wt[:, 300:].abs().max()) == 0.0
    bank, pooled = ops.imgbank_pool(dev(feat), wt, dev(bias), 300)
    ref = R.img_memory_bank(torch.from_numpy(feat), torch.from_numpy(w), torch.from_numpy(bias))
    assert H.relerr(bank.cpu(), ref) < 1e-5
    assert np.array_equal(pooled.cpu().numpy(), feat.max(axis=2))   # max is exact


def test_errors_are_loud():
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))
    with pytest.raises(RuntimeError, match="d_kv"):
        ops.sq_mha_core(torch.zeros(1, 64, device=DEV), torch.zeros(1, 4, 300, device=DEV), None, 1, 64,
                        torch.zeros(64, 300, device=DEV), None, torch.zeros(64, 300, device=DEV), None)


def test_bilstm_text_bank_against_golden_and_oracle():
    """HIP embedding+packed-BiLSTM vs the reference's get_text_memory_bank golden, then a bigger ragged batch
    (lengths 1..T, one at T) vs the oracle (torch CPU nn.LSTM)."""
    g = H.load_golden("text_bank.npz")
    V = int(g["V"])
    shapes = {k: s for k, s in H.surface().items() if k.startswith("lstm.")}
    shapes["embedding.weight"] = (V, 300)
    pc = H.params_for(shapes)
    p = dparams(pc)

    def weights():
        return [tuple(p["lstm.%s_l%d%s" % (n, l, sfx)] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
                for l in range(2) for sfx in ("", "_reverse")]

    bank = ops.bilstm(dev(g["tok"]), dev(g["lens"]), p["embedding.weight"], weights(), 150, 2)
    assert H.maxabs(bank.cpu(), g["bank"]) < 2e-6
    rs = np.random.RandomState(4)
    B, T = 37, 100
    lens = rs.randint(1, T + 1, size=B).astype(np.int64)
    lens[0], lens[1] = T, 1
    tok = np.zeros((B, T), np.int64)
    for b in range(B):
        tok[b, :lens[b]] = rs.randint(1, V, size=lens[b])
    ref = R.text_memory_bank(pc, torch.from_numpy(tok), torch.from_numpy(lens))
    bank = ops.bilstm(dev(tok), dev(lens), p["embedding.weight"], weights(), 150, 2).cpu()
    assert H.maxabs(bank, ref) < 5e-6
    for b in range(B):
        if lens[b] < T:
            assert float(bank[b, lens[b]:].abs().max()) == 0.0


def _bf16_round(x):
    return torch.as_tensor(x).to(torch.bfloat16).float()


@pytest.mark.parametrize("form", [32, 16, "packed"])
@pytest.mark.parametrize("Hn,tag,L,masked", GI.MHA_CASES)
def test_sq_mha_core_bf16(Hn, tag, L, masked, form):
    """bf16-operand kernels (32: v_mfma_f32_32x32x16_bf16, one workgroup per sample; "packed": the same with the masked
    bank's live rows packed by a plan; 16: the 16x16x32 form): (i) exact-logic check against the oracle fed the SAME
    bf16-rounded bank and K/V weights (only fp32 summation order differs -> tight), (ii) error vs the fp32 reference golden
    is reported and bounded loosely (bf16 operands: ~3 significant digits)."""
    packed = form == "packed"
    if packed and not masked:
        pytest.skip("a packing plan is for masked banks")
    form = 32 if packed else form
    g = H.load_golden("mha.npz")
    name = "h%d_%s" % (Hn, tag)
    pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
    p = dparams(pc)
    q, bank, mask = GI.mha_case(Hn, tag, L, masked)
    a = name + ".slf_attn."
    qh = ops.linear(dev(q), p[a + "w_qs.weight"], p[a + "w_qs.bias"])
    bank_bf = ops.cast_pad_bf16(dev(bank))
    assert bank_bf.shape == (bank.shape[0], L, ops.BANK_LD)
    assert torch.equal(bank_bf[..., :300].float().cpu(), _bf16_round(bank))      # RNE like torch
    assert float(bank_bf[..., 300:].float().abs().max()) == 0.0
    wp = ops.pack_kv_weights_bf16(p[a + "w_ks.weight"], p[a + "w_vs.weight"], Hn, 128, form=form)
    dmask = None if mask is None else dev(mask)
    o, attn = ops.sq_mha_core_bf16(qh, bank_bf, dmask, Hn, 128, wp, p[a + "w_ks.bias"], p[a + "w_vs.bias"],
                                   plan=ops.sq_mha_plan(dmask) if packed else None)
    # (i) oracle on bf16-rounded operands
    pr = dict(pc)
    pr[a + "w_ks.weight"] = _bf16_round(pc[a + "w_ks.weight"])
    pr[a + "w_vs.weight"] = _bf16_round(pc[a + "w_vs.weight"])
    tb = _bf16_round(bank)
    tm = None if mask is None else torch.from_numpy(mask)
    B = q.shape[0]
    qh_c = qh.cpu().view(B, Hn, 128)
    kh = torch.nn.functional.linear(tb, pr[a + "w_ks.weight"], pr[a + "w_ks.bias"]).view(B, L, Hn, 128)
    vh = torch.nn.functional.linear(tb, pr[a + "w_vs.weight"], pr[a + "w_vs.bias"]).view(B, L, Hn, 128)
    s = torch.einsum("bhd,blhd->bhl", qh_c, kh) / float(np.power(128, 0.5))
    if tm is not None:
        s = s.masked_fill(tm[:, None, :] == 0.0, float("-inf"))
    pa = torch.softmax(s, dim=2)
    o_ref = torch.einsum("bhl,blhd->bhd", pa, vh).reshape(B, Hn * 128)
    assert H.maxabs(attn.cpu(), pa.permute(1, 0, 2).reshape(Hn * B, 1, L)) < 2e-5
    assert H.relerr(o.cpu(), o_ref) < 2e-5
    # (ii) against the fp32 reference
    err = H.maxabs(attn.cpu(), g[name + "_attn"])
    assert err < 5e-2, err


@pytest.mark.parametrize("Hn", [1, 3, 8])
def test_sq_mha_core_bf16_random_batches_vs_fp32_core(Hn):
    """bf16 core vs the exact-fp32 core on random banks: odd head counts (a half-workgroup without a head), a batch
    that fills the chip (one workgroup owns all heads) and one that does not (head pairs split over workgroups)."""
    name = "h%d_img" % (Hn if Hn in (1, 4, 8) else 8)
    pc = H.params_for(H.mha_shapes(8 if Hn == 3 else Hn), prefix=name + ".")
    p = dparams(pc)
    a = name + ".slf_attn."
    wq, wk, wv = (p[a + k][:Hn * 128].contiguous() for k in ("w_qs.weight", "w_ks.weight", "w_vs.weight"))
    bq, bk, bv = (p[a + k][:Hn * 128].contiguous() for k in ("w_qs.bias", "w_ks.bias", "w_vs.bias"))
    wp = ops.pack_kv_weights_bf16(wk, wv, Hn, 128)
    for B, L in ((5, 196), (256, 196), (256, 37)):
        rs = np.random.RandomState(B + L + Hn)
        q = dev(rs.standard_normal((B, 300)).astype(np.float32))
        bank32 = dev(rs.standard_normal((B, L, 300)).astype(np.float32))
        qh = ops.linear(q, wq, bq)
        o, attn = ops.sq_mha_core_bf16(qh, ops.cast_pad_bf16(bank32), None, Hn, 128, wp, bk, bv)
        o32, attn32 = ops.sq_mha_core(qh, bank32, None, Hn, 128, wk, bk, wv, bv)
        assert torch.isfinite(attn).all() and torch.isfinite(o).all()
        assert H.maxabs(attn.cpu(), attn32.cpu()) < 5e-3, (B, L)
        assert H.maxabs(o.cpu(), o32.cpu()) < 2e-2, (B, L)


@pytest.mark.parametrize("Hn", [11, 16])
def test_sq_mha_core_bf16_more_heads_than_probability_rows(Hn):
    """A workgroup that owns more than eight heads (chip-filling batch, 11 / 16 heads) recycles the LDS rows of the head
    probabilities and the partial-score slots behind their guard counters (unit queue of csrc/sq_mha_bf16.hip); a masked
    batch with ragged lengths and one fully live row next to it; repeated launches give identical bits."""
    rs = np.random.RandomState(7 + Hn)
    wq, wk, wv = (dev((0.05 * rs.standard_normal((Hn * 128, 300))).astype(np.float32)) for _ in range(3))
    bq, bk, bv = (dev((0.05 * rs.standard_normal(Hn * 128)).astype(np.float32)) for _ in range(3))
    wp = ops.pack_kv_weights_bf16(wk, wv, Hn, 128)
    for B, L, masked in ((256, 40, False), (256, 100, True)):
        q = dev(rs.standard_normal((B, 300)).astype(np.float32))
        bank32 = dev(rs.standard_normal((B, L, 300)).astype(np.float32))
        mask = None
        if masked:
            lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, B))), 1, L).astype(int)
            lens[0] = L
            m = np.zeros((B, L), np.float32)
            for b in range(B):
                m[b, :lens[b]] = 1
            mask = dev(m)
        qh = ops.linear(q, wq, bq)
        bb = ops.cast_pad_bf16(bank32)
        o, attn = ops.sq_mha_core_bf16(qh, bb, mask, Hn, 128, wp, bk, bv)
        o2, attn2 = ops.sq_mha_core_bf16(qh, bb, mask, Hn, 128, wp, bk, bv)
        assert torch.equal(o, o2) and torch.equal(attn, attn2)
        o32, attn32 = ops.sq_mha_core(qh, bank32, mask, Hn, 128, wk, bk, wv, bv)
        assert torch.isfinite(attn).all() and torch.isfinite(o).all()
        assert H.maxabs(attn.cpu(), attn32.cpu()) < 5e-3, (B, L)
        assert H.maxabs(o.cpu(), o32.cpu()) < 2e-2, (B, L)


def _split_ref(qh, bank, mask, Hn, wk, wv, bv):
    """fp64 attention core (submodules.py:55-119, len_q == 1) on the fp32 operands -> (o [B, H*128], attn [H*B, 1, L])."""
    B, L, _ = bank.shape
    q64 = qh.double().view(B, Hn, 128)
    kh = (bank.double() @ wk.double().t()).view(B, L, Hn, 128)
    vh = (bank.double() @ wv.double().t() + bv.double()).view(B, L, Hn, 128)
    s = torch.einsum("bhd,blhd->bhl", q64, kh) / float(np.sqrt(128.0))
    if mask is not None:
        s = s.masked_fill(mask[:, None, :] == 0.0, float("-inf"))
    pa = torch.softmax(s, dim=2)
    o = torch.einsum("bhl,blhd->bhd", pa, vh).reshape(B, Hn * 128)
    return o, pa.permute(1, 0, 2).reshape(Hn * B, 1, L)


@pytest.mark.parametrize("Hn,tag,L,masked", GI.MHA_CASES)
def test_sq_mha_core_split_against_goldens(Hn, tag, L, masked):
    """Split-bf16 attention core (csrc/sq_mha_split_bf16.hip: hi + lo operands, three bf16 MFMAs per product, bank rows in two
    halves joined by an online-softmax merge) against the REFERENCE's goldens under the fp32 kernel's own bounds: returned
    attention <= 1e-5, layer output <= 2e-5 (submodules.py:55-119 -> moudles.py:207-230)."""
    g = H.load_golden("mha.npz")
    name = "h%d_%s" % (Hn, tag)
    pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
    p = dparams(pc)
    q, bank, mask = GI.mha_case(Hn, tag, L, masked)
    a = name + ".slf_attn."
    qh = ops.linear(dev(q), p[a + "w_qs.weight"], p[a + "w_qs.bias"])
    sp = ops.split_pad_bf16(dev(bank))
    assert sp.shape == (2,) + tuple(bank.shape[:2]) + (ops.BANK_LD,)
    hi, lo = sp[0][..., :300].float().cpu(), sp[1][..., :300].float().cpu()
    assert torch.equal(hi, _bf16_round(bank)) and torch.equal(lo, _bf16_round(torch.as_tensor(bank) - hi))
    assert float(sp[..., 300:].float().abs().max()) == 0.0
    wp = ops.pack_kv_weights_split(p[a + "w_ks.weight"], p[a + "w_vs.weight"], Hn, 128)
    dm = None if mask is None else dev(mask)
    o, attn = ops.sq_mha_core_split(qh, sp, dm, Hn, 128, wp, p[a + "w_ks.bias"], p[a + "w_vs.bias"])
    err_a = H.maxabs(attn.cpu(), g[name + "_attn"])
    o32, _ = ops.sq_mha_core(qh, dev(bank), dm, Hn, 128, p[a + "w_ks.weight"], p[a + "w_ks.bias"], p[a + "w_vs.weight"],
                             p[a + "w_vs.bias"])
    err_o = H.maxabs(o.cpu(), o32.cpu())
    y = ops.linear(o, p[a + "fc.weight"], p[a + "fc.bias"], residual=dev(q))
    y = ops.layernorm(y, p[a + "layer_norm.gamma"], p[a + "layer_norm.beta"])
    f = name + ".pos_ffn."
    h1 = ops.linear(y, p[f + "w_1.weight"].squeeze(-1).contiguous(), p[f + "w_1.bias"], act=ops.ACT_RELU)
    z = ops.linear(h1, p[f + "w_2.weight"].squeeze(-1).contiguous(), p[f + "w_2.bias"], residual=y)
    out = ops.layernorm(z, p[f + "layer_norm.gamma"], p[f + "layer_norm.beta"])
    err_y = H.maxabs(out.cpu(), g[name + "_out"])
    print("split core %s: attn %.2e, o vs exact-f32 core %.2e, layer out %.2e" % (name, err_a, err_o, err_y))
    assert err_a < 1e-5 and err_y < 2e-5 and err_o < 2e-5
    o2, none = ops.sq_mha_core_split(qh, sp, dm, Hn, 128, wp, p[a + "w_ks.bias"], p[a + "w_vs.bias"], want_attn=False)
    assert none is None and torch.equal(o2, o)


@pytest.mark.parametrize("Hn", [1, 3, 8, 16])
def test_sq_mha_core_split_random_batches_vs_fp64(Hn):
    """Split-bf16 core against fp64 on the fp32 operands: chip-filling batches (one workgroup owns every head pair) and small ones
    (head pairs split over workgroups; 16 heads always split: a workgroup owns at most eight), odd head counts, one half
    (L <= 112) and two halves (L = 113: one tile in the second; 196; 208), masks with ragged lengths, holes, a sample whose live
    rows all sit in the SECOND half (the first half's maximum is -inf) and one whose live rows all sit in the first; repeated
    launches give identical bits.  Bounds: hi + lo carry 16 mantissa bits (2^-17 per operand) against fp32's 24, so the split core
    sits ~15x above the exact-f32 core's own error against fp64 (printed next to it: 5e-6 / 3e-5 against 3e-7 / 2e-6 at B = 256,
    L = 196, |o| ~ 5) -- gated at 2e-5 on the probabilities and 1e-4 on o; the model-level gate is the 1e-4 on the logits."""
    rs = np.random.RandomState(100 + Hn)
    wq, wk, wv = (dev((0.06 * rs.standard_normal((Hn * 128, 300))).astype(np.float32)) for _ in range(3))
    bq, bk, bv = (dev((0.05 * rs.standard_normal(Hn * 128)).astype(np.float32)) for _ in range(3))
    wp = ops.pack_kv_weights_split(wk, wv, Hn, 128)
    for B, L, masked in ((5, 196, False), (256, 196, False), (256, 37, False), (7, 113, False), (3, 208, False), (2, 1, False),
                         (256, 100, True), (9, 196, True), (4, 112, True), (300, 128, True)):
        q = dev(rs.standard_normal((B, 300)).astype(np.float32))
        bank32 = dev((1.5 * rs.standard_normal((B, L, 300))).astype(np.float32))
        mask = None
        if masked:
            lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, B))), 1, L).astype(int)
            lens[0] = L
            lens[1] = 1
            m = np.zeros((B, L), np.float32)
            for b in range(B):
                m[b, :lens[b]] = 1
            if L > 120:
                m[2, :] = 0
                m[2, 115:L - 3] = 1                          # live rows in the second half only
                m[3, :] = 1
                m[3, 50:] = 0                               # ... in the first half only (whole batch still takes two halves)
            m[B - 1, ::3] = 0                               # holes
            m[B - 1, 0] = 1
            mask = dev(m)
        qh = ops.linear(q, wq, bq)
        sp = ops.split_pad_bf16(bank32)
        o, attn = ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wp, bk, bv)
        o2, attn2 = ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wp, bk, bv)
        assert torch.equal(o, o2) and torch.equal(attn, attn2)
        assert torch.isfinite(attn).all() and torch.isfinite(o).all()
        o64, a64 = _split_ref(qh.cpu(), bank32.cpu(), None if mask is None else mask.cpu(), Hn, wk.cpu(), wv.cpu(), bv.cpu())
        o32, a32 = ops.sq_mha_core(qh, bank32, mask, Hn, 128, wk, bk, wv, bv)
        ea, eo = H.maxabs(attn.cpu().double(), a64), H.maxabs(o.cpu().double(), o64)
        print("split core H=%d B=%d L=%d masked=%s: attn %.2e (f32 core %.2e), o %.2e (f32 core %.2e)"
              % (Hn, B, L, masked, ea, H.maxabs(a32.cpu().double(), a64), eo, H.maxabs(o32.cpu().double(), o64)))
        assert ea < 2e-5 and eo < 1e-4, (B, L, masked)
        if mask is not None:
            assert float(attn.view(Hn, B, L)[:, mask == 0].abs().max()) == 0.0       # masked positions: exactly 0


def test_sq_mha_core_split_fully_masked_sample_is_nan_like_the_reference():
    """softmax of an all -inf row is NaN in the reference (submodules.py:113-116) and so is bmm(attn, v): the split core returns NaN
    probabilities AND a NaN output row for such a sample, and the other samples of the batch are untouched."""
    rs = np.random.RandomState(5)
    Hn, B = 4, 6
    wq, wk, wv = (dev((0.06 * rs.standard_normal((Hn * 128, 300))).astype(np.float32)) for _ in range(3))
    bq, bk, bv = (dev((0.05 * rs.standard_normal(Hn * 128)).astype(np.float32)) for _ in range(3))
    wp = ops.pack_kv_weights_split(wk, wv, Hn, 128)
    for L in (100, 196):
        q = dev(rs.standard_normal((B, 300)).astype(np.float32))
        bank32 = dev(rs.standard_normal((B, L, 300)).astype(np.float32))
        m = np.ones((B, L), np.float32)
        m[2, :] = 0
        mask = dev(m)
        qh = ops.linear(q, wq, bq)
        o, attn = ops.sq_mha_core_split(qh, ops.split_pad_bf16(bank32), mask, Hn, 128, wp, bk, bv)
        o32, attn32 = ops.sq_mha_core(qh, bank32, mask, Hn, 128, wk, bk, wv, bv)
        # (the exact-f32 core skips the V pass of a sample without live rows: its attn is NaN like the reference's, its o is 0)
        assert torch.isnan(o[2]).all() and not torch.isnan(o[[0, 1, 3, 4, 5]]).any() and not torch.isnan(o32[[0, 1, 3, 4, 5]]).any()
        assert torch.equal(torch.isnan(attn), torch.isnan(attn32)) and torch.isnan(attn.view(Hn, B, L)[:, 2]).all()
        keep = [0, 1, 3, 4, 5]
        assert H.maxabs(o[keep].cpu(), o32[keep].cpu()) < 1e-4


@pytest.mark.parametrize("Hn,B,L", [(8, 256, 100), (8, 300, 112), (4, 64, 100), (1, 5, 37), (3, 33, 64), (8, 1, 100), (8, 256, 9)])
def test_sq_mha_core_split_grouped_plan_equals_one_workgroup_per_sample(Hn, B, L):
    """Masked banks through the GROUPED form of the split-bf16 core (the samples of a group -- whole 16-row tiles, <= 7 tiles and 7
    samples -- share a workgroup's staging and weight stream; scores against each row's own sample's query, softmax joined per sample,
    weighted sums flushed per sample) against the same core with one workgroup per sample (fp32 summation order only) and against
    fp64: MVSA-like lengths incl. 1 / 16 / 17 / 32 / L, a mask with holes, a mask whose live rows start late; the plan's groups
    cover the batch in order; repeated launches give identical bits."""
    rs = np.random.RandomState(Hn * 1000 + B + L)
    wq, wk, wv = (dev((0.06 * rs.standard_normal((Hn * 128, 300))).astype(np.float32)) for _ in range(3))
    bq, bk, bv = (dev((0.05 * rs.standard_normal(Hn * 128)).astype(np.float32)) for _ in range(3))
    wp = ops.pack_kv_weights_split(wk, wv, Hn, 128)
    q = dev(rs.standard_normal((B, 300)).astype(np.float32))
    bank32 = dev((1.5 * rs.standard_normal((B, L, 300))).astype(np.float32))
    lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, B))), 1, L).astype(int)
    lens[0] = L
    for i, v in enumerate((1, 16, 17, 32)):
        if i + 1 < B:
            lens[i + 1] = min(v, L)
    m = np.zeros((B, L), np.float32)
    for b in range(B):
        m[b, :lens[b]] = 1
    if B > 8:
        m[6, ::2] = 0
        m[6, 0] = 1                                           # holes
        m[7, :] = 0
        m[7, min(L - 1, 20):min(L, 30)] = 1                   # live rows that start late
    mask = dev(m)
    qh = ops.linear(q, wq, bq)
    sp = ops.split_pad_bf16(bank32)
    plan = ops.sq_mha_split_plan(mask)
    groups, off, lv = _decode_plan(plan, B)
    assert groups[0][0] == 0 and sum(g[1] for g in groups) == B and all(g[2] <= 112 and g[2] % 16 == 0 and g[1] <= 7 for g in groups)
    assert all(groups[i][0] + groups[i][1] == groups[i + 1][0] for i in range(len(groups) - 1))
    live = np.array([np.flatnonzero(m[b]).max() + 1 for b in range(B)])
    assert np.array_equal(lv, live) and (off % 16 == 0).all()
    o, none = ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wp, bk, bv, want_attn=False, plan=plan)
    o1, _ = ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wp, bk, bv, want_attn=False)
    o2, _ = ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wp, bk, bv, want_attn=False, plan=plan)
    assert none is None and torch.equal(o, o2) and torch.isfinite(o).all()
    o64, _ = _split_ref(qh.cpu(), bank32.cpu(), mask.cpu(), Hn, wk.cpu(), wv.cpu(), bv.cpu())
    e_same, e64 = H.maxabs(o.cpu(), o1.cpu()), H.maxabs(o.cpu().double(), o64)
    print("grouped split core H=%d B=%d L=%d: %d groups; vs per-sample %.2e, vs fp64 %.2e" % (Hn, B, L, len(groups), e_same, e64))
    assert e_same < 5e-6 and e64 < 1e-4
    with pytest.raises(ValueError):
        ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wp, bk, bv, want_attn=True, plan=plan)      # no attn output in the grouped form
    if B > 2:                                                  # a sample without a live row: NaN output row like the reference, others untouched
        m2 = m.copy()
        m2[2, :] = 0
        mk2 = dev(m2)
        o3, _ = ops.sq_mha_core_split(qh, sp, mk2, Hn, 128, wp, bk, bv, want_attn=False, plan=ops.sq_mha_split_plan(mk2))
        keep = [b for b in range(B) if b != 2]
        assert torch.isnan(o3[2]).all() and H.maxabs(o3[keep].cpu(), o1[keep].cpu()) < 5e-6


def test_a_plan_of_the_other_kind_is_refused_by_the_wrapper_and_by_the_kernel():
    """The packed bf16 kernel's plan (8-row alignment, up to 16 samples / 128 rows per group) and the grouped split-bf16 core's
    (16 / 7 / 112) have the SAME size for a batch: the wrappers take only the tensor their own plan builder returned, and a plan
    that reaches a kernel anyway (tag forged here) is recognised by its header word -- the launch writes nothing and raises the
    library's status word instead of indexing past its LDS maps."""
    from mgnns_amd import _lib
    rs = np.random.RandomState(5)
    Hn, B, L = 8, 64, 100
    wk, wv = (dev((0.06 * rs.standard_normal((Hn * 128, 300))).astype(np.float32)) for _ in range(2))
    bk, bv = (dev((0.05 * rs.standard_normal(Hn * 128)).astype(np.float32)) for _ in range(2))
    qh = dev(rs.standard_normal((B, Hn * 128)).astype(np.float32))
    bank32 = dev(rs.standard_normal((B, L, 300)).astype(np.float32))
    m = np.zeros((B, L), np.float32)
    for b in range(B):
        m[b, :1 + (b * 7) % 9] = 1                     # short samples: the packed plan puts 16 of them into a group
    mask = dev(m)
    packed, grouped = ops.sq_mha_plan(mask), ops.sq_mha_split_plan(mask)
    assert packed.numel() == grouped.numel() and int(packed[2]) != int(grouped[2]) and int(packed[1]) == int(grouped[1]) == B
    assert max(g[1] for g in _decode_plan(packed, B)[0]) > 7
    sp, wps = ops.split_pad_bf16(bank32), ops.pack_kv_weights_split(wk, wv, Hn, 128)
    bb, wp32 = ops.cast_pad_bf16(bank32), ops.pack_kv_weights_bf16(wk, wv, Hn, 128, form=32)
    with pytest.raises(ValueError, match="sq_mha_split_plan"):
        ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wps, bk, bv, want_attn=False, plan=packed)
    with pytest.raises(ValueError, match="sq_mha_split_plan"):
        ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wps, bk, bv, want_attn=False, plan=grouped.clone())     # (the tag does not survive)
    with pytest.raises(ValueError, match="sq_mha_plan"):
        ops.sq_mha_core_bf16(qh, bb, mask, Hn, 128, wp32, bk, bv, want_attn=False, plan=grouped)
    torch.cuda.synchronize()
    _lib.take_status()
    forged = packed.clone()
    forged._mg_plan_kind = 'grouped'
    ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wps, bk, bv, want_attn=False, plan=forged)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="status 3"):
        _lib.take_status()
    forged = grouped.clone()
    forged._mg_plan_kind = 'packed'
    ops.sq_mha_core_bf16(qh, bb, mask, Hn, 128, wp32, bk, bv, want_attn=False, plan=forged)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="status 3"):
        _lib.take_status()
    # and the real plans still run
    o, _ = ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wps, bk, bv, want_attn=False, plan=grouped)
    o1, _ = ops.sq_mha_core_split(qh, sp, mask, Hn, 128, wps, bk, bv, want_attn=False)
    torch.cuda.synchronize()
    _lib.take_status()
    assert H.maxabs(o.cpu(), o1.cpu()) < 5e-6


def _decode_plan(plan, B):
    pl = plan.cpu().numpy()
    ng = int(pl[0])
    groups = [tuple(int(x) for x in pl[4 + 4 * g: 4 + 4 * g + 3]) for g in range(ng)]
    off = pl[4 + 4 * B: 4 + 6 * B: 2].astype(int)
    lv = pl[4 + 4 * B + 1: 4 + 6 * B: 2].astype(int)
    return groups, off, lv


@pytest.mark.parametrize("Hn,B,L", [(8, 256, 100), (8, 300, 128), (4, 64, 100), (1, 5, 37), (3, 33, 64), (8, 1, 100), (8, 256, 9)])
def test_sq_mha_core_bf16_packed_plan_equals_one_workgroup_per_sample(Hn, B, L):
    """Masked banks with a packing plan (csrc/sq_mha32_bf16.hip: several short samples per workgroup, 8-row aligned, segmented
    softmax, block-wise weighted sums) against the same kernel family with one workgroup per sample, and against the exact-fp32
    core: MVSA-like ragged lengths, a full-length row, lengths 1 / 8 / 9, masks with HOLES (any 0/1 pattern is legal at the
    operator), a fully masked sample (NaN row, like the reference's softmax of -inf), odd head counts, batches beyond the grid;
    the plan itself: every sample exactly once, groups of <= 128 rows and <= 16 samples, 8-aligned offsets."""
    rs = np.random.RandomState(Hn * 1000 + B + L)
    wq, wk, wv = (dev((0.05 * rs.standard_normal((Hn * 128, 300))).astype(np.float32)) for _ in range(3))
    bq, bk, bv = (dev((0.05 * rs.standard_normal(Hn * 128)).astype(np.float32)) for _ in range(3))
    wp = ops.pack_kv_weights_bf16(wk, wv, Hn, 128, form=32)
    lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, B))), 1, L).astype(int)
    lens[0] = L
    for i, v in enumerate((1, 8, 9, 16, 17)):
        if i + 1 < B:
            lens[i + 1] = min(v, L)
    m = np.zeros((B, L), np.float32)
    for b in range(B):
        m[b, :lens[b]] = 1
        if lens[b] > 3 and b % 3 == 0:                       # holes (the last live position stays)
            m[b, rs.randint(0, lens[b] - 1, size=max(1, lens[b] // 4))] = 0
    dead = B - 1 if B > 7 else None
    if dead is not None:
        m[dead] = 0
    mask = dev(m)
    q = dev(rs.standard_normal((B, 300)).astype(np.float32))
    bank32 = dev(rs.standard_normal((B, L, 300)).astype(np.float32))
    qh = ops.linear(q, wq, bq)
    bb = ops.cast_pad_bf16(bank32)
    plan = ops.sq_mha_plan(mask)
    groups, off, lv = _decode_plan(plan, B)
    want_lv = np.array([0 if not m[b].any() else int(np.nonzero(m[b])[0][-1]) + 1 for b in range(B)])
    assert np.array_equal(lv, want_lv)
    seen = np.zeros(B, int)
    for first, cnt, rows in groups:
        assert 1 <= cnt <= 16 and rows <= 128
        r = 0
        for b in range(first, first + cnt):
            seen[b] += 1
            assert off[b] == r and r % 8 == 0
            r += max(8, (lv[b] + 7) // 8 * 8)
        assert r == rows
    assert (seen == 1).all()
    o, attn = ops.sq_mha_core_bf16(qh, bb, mask, Hn, 128, wp, bk, bv, plan=plan)
    o2, attn2 = ops.sq_mha_core_bf16(qh, bb, mask, Hn, 128, wp, bk, bv, plan=plan)
    o1, attn1 = ops.sq_mha_core_bf16(qh, bb, mask, Hn, 128, wp, bk, bv)                     # one workgroup per sample
    o32, attn32 = ops.sq_mha_core(qh, bank32, mask, Hn, 128, wk, bk, wv, bv)
    live = np.ones(B, bool)
    if dead is not None:
        live[dead] = False
        for t in (o, o1):
            assert torch.isnan(t[dead]).all()
        for t in (attn, attn1):
            assert torch.isnan(t.view(Hn, B, L)[:, dead]).all()
    lv_t = torch.from_numpy(live).to(o.device)
    assert torch.equal(o[lv_t], o2[lv_t]) and torch.equal(attn.view(Hn, B, L)[:, lv_t], attn2.view(Hn, B, L)[:, lv_t])
    assert torch.isfinite(o[lv_t]).all() and torch.isfinite(attn.view(Hn, B, L)[:, lv_t]).all()
    assert H.maxabs(attn.view(Hn, B, L)[:, lv_t].cpu(), attn1.view(Hn, B, L)[:, lv_t].cpu()) < 2e-6
    assert H.maxabs(o[lv_t].cpu(), o1[lv_t].cpu()) < 2e-5
    assert float(attn.view(Hn, B, L)[:, lv_t].cpu()[:, torch.from_numpy(m[live] == 0)].abs().max()) == 0.0    # masked positions: exactly 0
    assert H.maxabs(attn.view(Hn, B, L)[:, lv_t].cpu(), attn32.view(Hn, B, L)[:, lv_t].cpu()) < 5e-3
    assert H.maxabs(o[lv_t].cpu(), o32[lv_t].cpu()) < 2e-2


@pytest.mark.parametrize("form,B,P,K", [(1, 1, 196, 2048), (1, 3, 196, 2048), (2, 1, 196, 2048), (2, 3, 196, 2048),
                                        (1, 2, 208, 128), (1, 2, 64, 192), (1, 2, 16, 64), (2, 2, 108, 256), (0, 131, 196, 256)])
def test_imgbank_pool_bf16(form, B, P, K):
    """bf16-operand bank kernels (form 1: one workgroup per sample, the map through an LDS-DMA ring; form 2: two workgroups per
    sample; form 0: chosen by the batch -- 131 samples are more than half a chip, i.e. the stream form) vs fp64 math on the SAME
    bf16-rounded operands (tight), exact fp32 max-pool, zero padding of the 320-wide bank rows, and the error vs the fp32 result
    (loose, reported).  Region counts down to one row tile and up to the 13-tile limit, K down to one trip of the main loop."""
    rs = np.random.RandomState(30 + B + P)
    feat = np.maximum(rs.standard_normal((B, K, P)), 0).astype(np.float32)
    feat[0, 5, :] = -rs.uniform(0.1, 1.0, P).astype(np.float32)          # an all-negative feature row (max < 0)
    w = (0.05 * rs.standard_normal((300, K))).astype(np.float32)
    bias = (0.05 * rs.standard_normal(300)).astype(np.float32)
    wp = ops.pack_imgbank_weights_bf16(dev(w))
    ops.imgbank_set_form(form)
    try:
        bank, pooled = ops.imgbank_pool_bf16(dev(feat), wp, dev(bias), 300)
        _, halves = ops.imgbank_pool_bf16(dev(feat), wp, dev(bias), 300, combine=False)
    finally:
        ops.imgbank_set_form(0)
    assert bank.shape == (B, P, ops.BANK_LD) and bank.dtype == torch.bfloat16
    assert np.array_equal(pooled.cpu().numpy(), feat.max(axis=2))
    assert np.array_equal(halves.max(dim=1).values.cpu().numpy(), feat.max(axis=2))
    assert float(bank[..., 300:].float().abs().max()) == 0.0
    fr = _bf16_round(feat).double()
    wr = _bf16_round(w).double()
    ref = torch.einsum("bkp,nk->bpn", fr, wr) + torch.from_numpy(bias).double()
    got = bank[..., :300].float().cpu().double()
    # the result itself is rounded to bf16 on store: half an ulp = 2^-9 relative
    assert float(((got - ref).abs() / (ref.abs() + 1e-2)).max()) < 6e-3
    if B <= 3 and K == 2048:                 # (the oracle's restatement is written for the model's 2048 channels)
        ref32 = R.img_memory_bank(torch.from_numpy(feat), torch.from_numpy(w), torch.from_numpy(bias))
        print("bf16 bank max abs err vs fp32: %.3e (|bank| max %.2f)" % (H.maxabs(got, ref32), float(ref32.abs().max())))


@pytest.mark.parametrize("B,D,NL", [(1, 300, 3), (37, 300, 3), (256, 300, 3), (9, 512, 7)])
def test_classifier_head_as_four_shares_equals_the_single_launch(B, D, NL):
    """mgnns_classifier_part_fwd: the four shares launched from four streams in any order complete the same logits as the
    one-launch classifier (fp32 association differs: 1e-5), launch after launch on the same buffers (the last share re-arms
    the arrival counter), and bit-identically whichever share lands last."""
    rs = np.random.RandomState(B + D)
    feats = [dev(rs.standard_normal((B, D)).astype(np.float32)) for _ in range(4)]
    w = dev((0.05 * rs.standard_normal((NL, 4 * D))).astype(np.float32))
    bias = dev(rs.standard_normal(NL).astype(np.float32))
    ref = ops.classifier_head(feats, w, bias)
    state = ops.classifier_head_state(B, NL, 4, w.device)
    streams = [torch.cuda.Stream() for _ in range(4)]
    first = None
    for order in ([0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 3, 0, 2], [0, 1, 2, 3]):
        torch.cuda.synchronize()
        state[2].fill_(float("nan"))
        torch.cuda.synchronize()
        for p in order:
            with torch.cuda.stream(streams[p]):
                ops.classifier_head_part(feats[p], p, 4, w, bias, state)
        torch.cuda.synchronize()
        assert int(state[1].item()) == 0
        assert H.maxabs(state[2].cpu(), ref.cpu()) < 1e-5
        if first is None:
            first = state[2].clone()
        assert torch.equal(state[2], first)


def test_imgbank_forms_agree_at_full_batch_and_repeat():
    """configs[2]'s bank shape (256 x 2048 x 196): the stream form (counted waits on two in-order request streams per workgroup)
    gives the same bits launch after launch with every compute unit busy, and agrees with the pair form (same products, same
    fp32 accumulation order per output) to bf16 rounding; pooled maxima identical."""
    g = torch.Generator(device=DEV).manual_seed(11)
    feat = torch.relu(torch.randn(256, 2048, 196, device=DEV, generator=g))
    w = torch.randn(300, 2048, device=DEV, generator=g) * 0.05
    bias = torch.randn(300, device=DEV, generator=g) * 0.05
    wp = ops.pack_imgbank_weights_bf16(w)
    try:
        ops.imgbank_set_form(1)
        bank1, pooled1 = ops.imgbank_pool_bf16(feat, wp, bias, 300)
        for _ in range(10):
            bank, pooled = ops.imgbank_pool_bf16(feat, wp, bias, 300)
            assert torch.equal(bank, bank1) and torch.equal(pooled, pooled1)
        ops.imgbank_set_form(2)
        bank2, pooled2 = ops.imgbank_pool_bf16(feat, wp, bias, 300)
    finally:
        ops.imgbank_set_form(0)
    assert torch.equal(pooled1, pooled2) and torch.equal(pooled1, feat.amax(dim=2))
    d = (bank1.float() - bank2.float()).abs()
    assert float((d / (bank2.float().abs() + 1e-2)).max()) < 1.2e-2          # at most one bf16 ulp apart


@pytest.mark.parametrize("Hn", [1, 4, 8])
def test_run_stack_fused_tail_matches_layer_by_layer_and_golden(Hn):
    """fusion.run_stack (fused core + fused tail + chained query projection) == the per-layer module path ==
    the reference layer golden (single layer), for a 2-layer stack vs the oracle."""
    from mgnns_amd.fusion import MyMultiHeadAttention, run_stack
    g = H.load_golden("mha.npz")
    for tag, L, masked in (("text", 100, True), ("img", 196, False)):
        name = "h%d_%s" % (Hn, tag)
        pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
        layer = MyMultiHeadAttention(Hn, 300, 128, dropout=0.5, need_mask=masked).eval()
        layer.load_state_dict({k[len(name) + 1:]: v for k, v in pc.items()})
        layer = layer.to(DEV)
        q, bank, mask = GI.mha_case(Hn, tag, L, masked)
        dm = None if mask is None else dev(mask)
        out1 = run_stack([layer], dev(q), dev(bank), dm)
        assert H.maxabs(out1.cpu(), g[name + "_out"]) < 2e-5
        dbank = dev(bank)
        out_mod, _ = layer(q=dev(q), k=dbank, v=dbank, mask=dm)
        assert H.maxabs(out1.cpu(), out_mod.cpu()) < 2e-5
        # two layers chained (same weights twice): oracle mha_stack semantics
        ref = R.mha_stack({("s.%d." % i) + k[len(name) + 1:]: v for i in range(2) for k, v in pc.items()}, "s",
                          torch.from_numpy(q), torch.from_numpy(bank), None if mask is None else torch.from_numpy(mask),
                          Hn, 128, 2)
        out2 = run_stack([layer, layer], dev(q), dev(bank), dm)
        assert H.maxabs(out2.cpu(), ref) < 3e-5


@pytest.mark.parametrize("Hn,tag,L,masked", GI.MHA_CASES)
def test_sq_mha_folded_bf16_against_goldens(Hn, tag, L, masked):
    """The composed-map folded attention on the bf16 matrix pipe (sq_mha_folded_bf16.hip): attention weights against the
    reference's golden (bf16 operands: 5e-3 absolute on probabilities), the head outputs rebuilt from its weighted bank rows
    against the exact fp32 core, and the weighted rows exactly against fp64 on the kernel's own probabilities and bf16 bank."""
    g = H.load_golden("mha.npz")
    name = "h%d_%s" % (Hn, tag)
    p = dparams(H.params_for(H.mha_shapes(Hn), prefix=name + "."))
    q, bank, mask = GI.mha_case(Hn, tag, L, masked)
    dm = None if mask is None else dev(mask)
    a = name + ".slf_attn."
    qh = ops.linear(dev(q), p[a + "w_qs.weight"], p[a + "w_qs.bias"])
    B = q.shape[0]
    wk = p[a + "w_ks.weight"].double().view(Hn, 128, 300)
    u = torch.einsum("hdf,bhd->bhf", wk, qh.double().view(B, Hn, 128)).float().reshape(B, Hn * 300).contiguous()
    bank_bf = ops.cast_pad_bf16(dev(bank))
    c, attn = ops.sq_mha_folded_bf16(u, bank_bf, dm, Hn, 128)
    ldc = (Hn * 300 + 31) // 32 * 32
    assert c.shape == (B, ldc) and attn.shape == (Hn * B, 1, L) and not c[:, Hn * 300:].any()
    c = c[:, :Hn * 300]
    assert H.maxabs(attn.cpu(), g[name + "_attn"]) < 5e-3
    c2, none = ops.sq_mha_folded_bf16(u, bank_bf, dm, Hn, 128, want_attn=False)
    assert none is None and torch.equal(c2[:, :Hn * 300], c)
    # weighted bank rows: fp64 on the kernel's probabilities (rounded to bf16 as the kernel feeds them) and the bf16 bank
    pr = attn.view(Hn, B, L).to(torch.bfloat16).double()
    want = torch.einsum("hbl,blf->bhf", pr, bank_bf[..., :300].double())
    got = c.double().reshape(B, Hn, 300)
    assert float((got - want).abs().max() / want.abs().max()) < 6e-3          # bf16 rounding of the stored result
    # head outputs W_v c + b_v against the exact fp32 core
    wv = p[a + "w_vs.weight"].double().view(Hn, 128, 300)
    o = (torch.einsum("hdf,bhf->bhd", wv, got) + p[a + "w_vs.bias"].double().view(1, Hn, 128)).reshape(B, Hn * 128)
    o_f, _ = ops.sq_mha_core(qh, dev(bank), dm, Hn, 128, p[a + "w_ks.weight"], p[a + "w_ks.bias"],
                             p[a + "w_vs.weight"], p[a + "w_vs.bias"])
    assert H.relerr(o.cpu(), o_f.cpu()) < 2e-2


def test_sq_mha_folded_bf16_ragged_lengths_and_masked_tiles():
    """L not a multiple of 16 or 32, a single row, the 208-row maximum, masks that blank whole row tiles (never staged),
    fewer heads than waves: against the exact fp32 folded kernel on the bf16-rounded bank."""
    rs = np.random.RandomState(78)
    for L, Hn, B in ((1, 8, 3), (17, 4, 2), (100, 8, 5), (196, 8, 4), (208, 1, 2), (33, 3, 2)):
        u = dev((0.3 * rs.standard_normal((B, Hn * 300))).astype(np.float32))
        bank = dev(rs.standard_normal((B, L, 300)).astype(np.float32))
        bank_bf = ops.cast_pad_bf16(bank)
        mask = np.ones((B, L), np.float32)
        for b in range(B):
            mask[b, rs.randint(1, L + 1):] = 0.0
        mask[0, :] = 1.0
        if L > 40:
            mask[1, 5:37] = 0.0                               # holes in front of live rows
        dm = dev(mask)
        c, attn = ops.sq_mha_folded_bf16(u, bank_bf, dm, Hn, 128)
        xb = bank_bf[..., :300].double()
        ub = u.to(torch.bfloat16).double().view(B, Hn, 300)
        sc = torch.einsum("bhf,blf->bhl", ub, xb) / (128 ** 0.5)
        sc = sc.masked_fill(dm.double()[:, None, :] == 0, float("-inf"))
        pr = torch.softmax(sc, dim=-1)                                           # [B, H, L]
        assert H.maxabs(attn.view(Hn, B, L).permute(1, 0, 2).cpu(), pr.cpu()) < 2e-5, L
        want = torch.einsum("bhl,blf->bhf", pr.to(torch.bfloat16).double(), xb)
        err = float((c[:, :Hn * 300].double().reshape(B, Hn, 300) - want).abs().max() / want.abs().max())
        assert err < 8e-3, (L, err)


@pytest.mark.parametrize("Hn,B", [(8, 256), (8, 37), (4, 16), (1, 5)])
def test_mha_tail_c16_cluster_forms_agree_and_match_fp64(Hn, B):
    """The tail behind the folded attention: one workgroup per tile, clusters of 2 / 4 / 8 that recompute the front, clusters that
    split the K of the composed output map and exchange partial sums -- against fp64 on the bf16-rounded operands of the first
    product (the later products round their own inputs: bf16-class bound), against each other (same value up to the summation
    order of the first product) and launch after launch (bit-equal: the exchange sums in rank order)."""
    g = torch.Generator(device=DEV).manual_seed(11 + Hn)
    r = lambda *shape: torch.randn(*shape, device=DEV, generator=g) * 0.05
    HD = Hn * 300
    ldc = (HD + 31) // 32 * 32
    c = torch.zeros(B, ldc, device=DEV, dtype=torch.bfloat16)
    c[:, :HD] = torch.randn(B, HD, device=DEV, generator=g).to(torch.bfloat16)
    q = torch.randn(B, 300, device=DEV, generator=g)
    fc, w1, w2, wq = r(300, HD), r(300, 300), r(300, 300), r(HD, 300)
    pk = {"fc_b": r(300), "g1": r(300) + 1, "be1": r(300), "b1": r(300), "b2": r(300), "g2": r(300) + 1, "be2": r(300),
          "fc": ops.pack_weight_bf16_split(fc), "w1": ops.pack_weight_bf16_split(w1), "w2": ops.pack_weight_bf16_split(w2)}
    bq = r(HD)
    nx = (ops.pack_weight_bf16_split(wq), bq, HD)

    def ln(x, gamma, beta):
        mu = x.mean(-1, keepdim=True)
        sd = (x - mu).pow(2).sum(-1, keepdim=True).div(x.shape[-1] - 1).sqrt()
        return gamma.double() * (x - mu) / (sd + 1e-6) + beta.double()
    bfr = lambda t: t.to(torch.bfloat16).double()
    y = ln(c[:, :HD].double() @ bfr(fc).t() + pk["fc_b"].double() + q.double(), pk["g1"], pk["be1"])
    h = torch.relu(bfr(y.float()) @ bfr(w1).t() + pk["b1"].double())
    want = ln(bfr(h.float()) @ bfr(w2).t() + pk["b2"].double() + y, pk["g2"], pk["be2"])
    want_u = bfr(want.float()) @ bfr(wq).t() + bq.double()
    ref = None
    for cluster, ksplit in ((1, False), (2, False), (4, False), (0, True), (2, True), (4, True), (8, True)):
        out, u = ops.mha_tail_c16(c, q, pk, 1e-6, nx, cluster=cluster, ksplit=ksplit)
        out2, u2 = ops.mha_tail_c16(c, q, pk, 1e-6, nx, cluster=cluster, ksplit=ksplit)
        assert torch.equal(out, out2) and torch.equal(u, u2), (cluster, ksplit)
        assert H.maxabs(out.cpu(), want.cpu()) < 2e-2 and H.relerr(u.cpu(), want_u.cpu()) < 2e-2, (cluster, ksplit)
        if ref is None:
            ref = (out, u)
        else:
            # (a different summation order of the first product can move an intermediate across a bf16 rounding boundary)
            assert H.maxabs(out.cpu(), ref[0].cpu()) < 1.2e-2 and H.relerr(u.cpu(), ref[1].cpu()) < 1.2e-2, (cluster, ksplit)
        last, none = ops.mha_tail_c16(c, q, pk, 1e-6, None, cluster=cluster, ksplit=ksplit)
        assert none is None and torch.equal(last, out), (cluster, ksplit)


@pytest.mark.parametrize("Hn", [1, 4, 8])
def test_run_stack_folded_bf16_matches_golden_and_the_explicit_bf16_stack(Hn):
    """bf16 mode + folded attention through fusion.run_stack (composed query map, one-bank-read attention, tail with the
    composed output map): the reference layer golden and a two-layer stack against the oracle at bf16 tolerance, and as close to
    them as the explicit bf16 stack is."""
    from mgnns_amd.fusion import MyMultiHeadAttention, MemoryBank, run_stack
    g = H.load_golden("mha.npz")
    for tag, L, masked in (("text", 100, True), ("img", 196, False)):
        name = "h%d_%s" % (Hn, tag)
        pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
        layer = MyMultiHeadAttention(Hn, 300, 128, dropout=0.5, need_mask=masked).eval()
        layer.load_state_dict({k[len(name) + 1:]: v for k, v in pc.items()})
        layer = layer.to(DEV)
        layer.slf_attn.precision = 'bf16'
        q, bank, mask = GI.mha_case(Hn, tag, L, masked)
        dm = None if mask is None else dev(mask)
        ref2 = R.mha_stack({("s.%d." % i) + k[len(name) + 1:]: v for i in range(2) for k, v in pc.items()}, "s",
                           torch.from_numpy(q), torch.from_numpy(bank), None if mask is None else torch.from_numpy(mask),
                           Hn, 128, 2)
        errs = {}
        for att in ("faithful", "folded"):
            layer.slf_attn.attention = att
            mb = MemoryBank(f32=dev(bank))
            out1 = run_stack([layer], dev(q), mb, dm)
            out2 = run_stack([layer, layer], dev(q), mb, dm)
            errs[att] = (H.maxabs(out1.cpu(), g[name + "_out"]), H.maxabs(out2.cpu(), ref2))
        print("stack h%d %s: explicit bf16 %.2e / %.2e, folded bf16 %.2e / %.2e" % ((Hn, tag) + errs["faithful"] + errs["folded"]))
        assert errs["folded"][0] < 3e-2 and errs["folded"][1] < 4e-2
        assert errs["folded"][0] < 2.5 * errs["faithful"][0] + 2e-3 and errs["folded"][1] < 2.5 * errs["faithful"][1] + 2e-3


@pytest.mark.parametrize("Hn", [4, 8])
def test_is_regu_head_difference_against_the_reference_golden(Hn):
    """MyMultiHeadAttention(is_regu=True) returns the head-difference term as a third value (moudles.py:220-229,
    submodules.py:38-52, 84-93): the reference's own output (golden), every attention variant; the operator alone against
    the oracle incl. a zero head (F.normalize's eps) and one head (0 / 0 = NaN, like the reference)."""
    from mgnns_amd.fusion import MyMultiHeadAttention
    g = H.load_golden("mha.npz")
    for tag, L, masked in (("text", 100, True), ("img", 196, False)):
        name = "h%d_%s" % (Hn, tag)
        pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
        layer = MyMultiHeadAttention(Hn, 300, 128, dropout=0.5, need_mask=masked, is_regu=True).eval()
        layer.load_state_dict({k[len(name) + 1:]: v for k, v in pc.items()})
        layer = layer.to(DEV)
        q, bank, mask = GI.mha_case(Hn, tag, L, masked)
        dm = None if mask is None else dev(mask)
        dbank = dev(bank)
        out, attn, hd = layer(q=dev(q), k=dbank, v=dbank, mask=dm)
        assert hd.shape == (q.shape[0],)
        assert H.maxabs(out.cpu(), g[name + "_out"]) < 2e-5 and H.maxabs(attn.cpu(), g[name + "_attn"]) < 1e-5
        assert H.maxabs(hd.cpu(), g[name + "_head_diff"]) < 1e-6
        layer.slf_attn.attention = 'folded'
        hd_f = layer(q=dev(q), k=dbank, v=dbank, mask=dm)[2]
        assert H.maxabs(hd_f.cpu(), g[name + "_head_diff"]) < 1e-6
        layer.slf_attn.attention = 'faithful'
    rs = np.random.RandomState(3 + Hn)
    o = rs.standard_normal((9, Hn, 128)).astype(np.float32)
    o[2, 1] = 0.0                                          # a zero head vector: normalised to 0, not NaN
    got = ops.head_diff(dev(o.reshape(9, -1)), Hn).cpu()
    assert H.maxabs(got, R.head_diff(torch.from_numpy(o))) < 1e-6
    assert torch.isnan(ops.head_diff(dev(o[:, 0].copy()), 1)).all()


@pytest.mark.parametrize("Hn", [4, 8])
def test_mha_tail_bf16_split_matches_fp32_tail(Hn):
    """split-bf16 fused tail (3 MFMA terms) vs the exact-fp32 fused tail on the same inputs: fp32-class agreement;
    the 1-term (plain bf16) tail only bf16-class."""
    name = "h%d_img" % Hn
    pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
    p = dparams(pc)
    rs = np.random.RandomState(5 + Hn)
    B = 37
    o = dev(rs.standard_normal((B, Hn * 128)).astype(np.float32))
    q = dev(rs.standard_normal((B, 300)).astype(np.float32))
    a, f = name + ".slf_attn.", name + ".pos_ffn."
    w1 = p[f + "w_1.weight"].squeeze(-1).contiguous()
    w2 = p[f + "w_2.weight"].squeeze(-1).contiguous()
    common = {"fc_b": p[a + "fc.bias"], "g1": p[a + "layer_norm.gamma"], "be1": p[a + "layer_norm.beta"],
              "b1": p[f + "w_1.bias"], "b2": p[f + "w_2.bias"], "g2": p[f + "layer_norm.gamma"],
              "be2": p[f + "layer_norm.beta"]}
    pk32 = dict(common, fc_wp=ops.pack_weight_f32(p[a + "fc.weight"]), w1_wp=ops.pack_weight_f32(w1),
                w2_wp=ops.pack_weight_f32(w2))
    nx32 = (ops.pack_weight_f32(p[a + "w_qs.weight"]), p[a + "w_qs.bias"], Hn * 128)
    out32, qh32 = ops.mha_tail(o, q, pk32, 1e-6, nx32)
    pkbf = dict(common, fc=ops.pack_weight_bf16_split(p[a + "fc.weight"]), w1=ops.pack_weight_bf16_split(w1),
                w2=ops.pack_weight_bf16_split(w2))
    nxbf = (ops.pack_weight_bf16_split(p[a + "w_qs.weight"]), p[a + "w_qs.bias"], Hn * 128)
    out3, qh3 = ops.mha_tail_bf16(o, q, pkbf, 1e-6, nxbf, terms=3)
    assert H.maxabs(out3.cpu(), out32.cpu()) < 5e-5
    assert H.relerr(qh3.cpu(), qh32.cpu()) < 5e-5
    out1, qh1 = ops.mha_tail_bf16(o, q, pkbf, 1e-6, nxbf, terms=1)
    e1 = H.maxabs(out1.cpu(), out32.cpu())
    assert 1e-4 < e1 < 5e-2
    # the split tail with fc's K split over a cluster (round 5; the next query through the exact-fp32 GEMM): fp32-class as well,
    # every cluster size, batches that are not a multiple of the tile, repeated launches on the same scratch
    nlin = (p[a + "w_qs.weight"], p[a + "w_qs.bias"])
    for Bk in (37, 256, 1):
        ok_, qk_ = o[:Bk].contiguous() if Bk <= B else dev(rs.standard_normal((Bk, Hn * 128)).astype(np.float32)), None
        qq = q[:Bk].contiguous() if Bk <= B else dev(rs.standard_normal((Bk, 300)).astype(np.float32))
        r_out, r_qh = ops.mha_tail(ok_, qq, pk32, 1e-6, nx32)
        for cl in (0, 2, 8):
            for rep in range(2):
                outk, qhk = ops.mha_tail_bf16(ok_, qq, pkbf, 1e-6, nxbf, terms=3, ksplit=True, cluster=cl, next_linear=nlin)
                assert H.maxabs(outk.cpu(), r_out.cpu()) < 5e-5 and H.relerr(qhk.cpu(), r_qh.cpu()) < 5e-5, (Bk, cl, rep)
        outn, qhn = ops.mha_tail_bf16(ok_, qq, pkbf, 1e-6, None, terms=3, ksplit=True)
        assert qhn is None and H.maxabs(outn.cpu(), r_out.cpu()) < 5e-5


@pytest.mark.parametrize("Hn", [1, 4, 8])
def test_mha_tail_bf16_k_split_over_the_cluster(Hn):
    """The explicit bf16 tail with fc's K = n_head * d_v split over a tile's cluster of workgroups (the last arriver adds the
    partial sums and finishes the tile, the next layer's w_qs as a second launch: mgnns_mha_tail_bf16_fwd with exchange buffers)
    against the one-workgroup-streams-all form: same chain, partial sums in another order -- an fp32 rounding difference that the
    chain's bf16 operand roundings (LayerNorm 1's output feeds w_1 as bf16) can turn into one bf16 ulp of an intermediate; batches
    that are not a multiple of the tile, every cluster size, repeated launches on the same scratch (the counters re-arm
    themselves), and the exact-logic check: with ONE rank nothing is split and the bits are the reference's."""
    name = "h%d_img" % Hn
    pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
    p = dparams(pc)
    a, f = name + ".slf_attn.", name + ".pos_ffn."
    w1 = p[f + "w_1.weight"].squeeze(-1).contiguous()
    w2 = p[f + "w_2.weight"].squeeze(-1).contiguous()
    pk = {"fc_b": p[a + "fc.bias"], "g1": p[a + "layer_norm.gamma"], "be1": p[a + "layer_norm.beta"], "b1": p[f + "w_1.bias"],
          "b2": p[f + "w_2.bias"], "g2": p[f + "layer_norm.gamma"], "be2": p[f + "layer_norm.beta"],
          "fc": ops.pack_weight_bf16_split(p[a + "fc.weight"]), "w1": ops.pack_weight_bf16_split(w1), "w2": ops.pack_weight_bf16_split(w2)}
    nx = (ops.pack_weight_bf16_split(p[a + "w_qs.weight"]), p[a + "w_qs.bias"], Hn * 128)
    rs = np.random.RandomState(Hn)
    for B in (256, 37, 1, 300):
        o = dev(rs.standard_normal((B, Hn * 128)).astype(np.float32))
        q = dev(rs.standard_normal((B, 300)).astype(np.float32))
        for nxt in (nx, None):
            ref_out, ref_qh = ops.mha_tail_bf16(o, q, pk, 1e-6, nxt, terms=1, ksplit=False)
            for cl in (0, 2, 3, 8):
                for rep in range(2):
                    out, qh = ops.mha_tail_bf16(o, q, pk, 1e-6, nxt, terms=1, ksplit=True, cluster=cl)
                    assert H.maxabs(out.cpu(), ref_out.cpu()) < 1e-2 and H.relerr(out.cpu(), ref_out.cpu()) < 1e-3, (B, cl, rep)
                    assert (qh is None) == (nxt is None)
                    if qh is not None:                       # (the projection rounds `out` to bf16 first: a 1e-6 change can flip a rounding)
                        assert H.maxabs(qh.cpu(), ref_qh.cpu()) < 1e-2 and H.relerr(qh.cpu(), ref_qh.cpu()) < 2e-3, (B, cl, rep)
    ws = pk["_cluster_ws_ks"]
    assert all(int(v[2].abs().sum()) == 0 for v in ws.values())               # the arrival counters are back at zero


@pytest.mark.parametrize("Hn,L,masked", [(8, 196, False), (8, 100, True), (4, 196, False), (1, 50, True)])
def test_fused_layer_bf16_equals_core_plus_tail(Hn, L, masked):
    """mgnns_sq_mha_layer_bf16_fwd (attention core + the tile's tail run by its last-finishing core workgroup, hand-over of
    `o` through system-scope stores and one relaxed atomic per workgroup) == the two separate launches, BIT FOR BIT, for
    batches that fill the chip (one workgroup per sample), smaller ones (head pairs split over workgroups), batches that are
    not a multiple of the 16-sample tile, repeated launches on the same counters, with and without a next-layer projection."""
    name = "h%d_img" % Hn
    pc = H.params_for(H.mha_shapes(Hn), prefix=name + ".")
    p = dparams(pc)
    a, f = name + ".slf_attn.", name + ".pos_ffn."
    w1 = p[f + "w_1.weight"].squeeze(-1).contiguous()
    w2 = p[f + "w_2.weight"].squeeze(-1).contiguous()
    pk = {"fc_b": p[a + "fc.bias"], "g1": p[a + "layer_norm.gamma"], "be1": p[a + "layer_norm.beta"], "b1": p[f + "w_1.bias"],
          "b2": p[f + "w_2.bias"], "g2": p[f + "layer_norm.gamma"], "be2": p[f + "layer_norm.beta"],
          "fc": ops.pack_weight_bf16_split(p[a + "fc.weight"]), "w1": ops.pack_weight_bf16_split(w1), "w2": ops.pack_weight_bf16_split(w2)}
    nx = (ops.pack_weight_bf16_split(p[a + "w_qs.weight"]), p[a + "w_qs.bias"], Hn * 128)
    wp = ops.pack_kv_weights_bf16(p[a + "w_ks.weight"], p[a + "w_vs.weight"], Hn, 128, form=16)      # (the fused layer runs the 16x16x32 core)
    counters = torch.zeros(64, dtype=torch.int32, device=DEV)
    rs = np.random.RandomState(Hn + L)
    for B in (256, 37, 16, 1, 300):
        bank = ops.cast_pad_bf16(dev(rs.standard_normal((B, L, 300)).astype(np.float32)))
        qh = dev(rs.standard_normal((B, Hn * 128)).astype(np.float32))
        q = dev(rs.standard_normal((B, 300)).astype(np.float32))
        mask = None
        if masked:
            m = np.ones((B, L), np.float32)
            for b in range(B):
                m[b, rs.randint(1, L + 1):] = 0.0
            mask = dev(m)
        o, _ = ops.sq_mha_core_bf16(qh, bank, mask, Hn, 128, wp, p[a + "w_ks.bias"], p[a + "w_vs.bias"], want_attn=False)
        for nxt in (nx, None):
            ref_out, ref_qh = ops.mha_tail_bf16(o, q, pk, 1e-6, nxt, terms=1, ksplit=False)      # (the fused layer runs the unsplit chain)
            for rep in range(3):
                out, qhn = ops.sq_mha_layer_bf16(qh, bank, mask, Hn, 128, wp, p[a + "w_ks.bias"], p[a + "w_vs.bias"], q, pk, 1e-6,
                                                 counters, nxt)
                torch.cuda.synchronize()
                assert torch.equal(out, ref_out), (B, rep, float((out - ref_out).abs().max()))
                assert (qhn is None) == (nxt is None) and (nxt is None or torch.equal(qhn, ref_qh)), (B, rep)
                assert int(counters.abs().sum()) == 0                 # every tile was handed over exactly once


def test_metrics_tail_softmax_argmax_confusion():
    """The evaluation tail on the device (ENGINE:828-838): softmax == torch.softmax, pred == argmax(softmax) incl. ties
    (first maximum), confusion matrix accumulated over batches, scores == sklearn's on the concatenated predictions."""
    from sklearn.metrics import accuracy_score, f1_score
    from mgnns_amd.metrics import Metrics, predict
    rs = np.random.RandomState(3)
    for NL in (3, 7):
        met = Metrics(NL, DEV)
        ys, ps = [], []
        for B in (1, 255, 600):
            logits = (3.0 * rs.standard_normal((B, NL))).astype(np.float32)
            logits[0, :] = 1.25                                       # an exact tie: the first class wins
            y = rs.randint(0, NL, size=B).astype(np.int64)
            probs, pred = met.update(dev(logits), dev(y), want_probs=True)
            ref = torch.softmax(torch.from_numpy(logits), dim=1)
            assert H.maxabs(probs.cpu(), ref) < 1e-6
            assert np.array_equal(pred.cpu().numpy(), ref.argmax(dim=1).numpy().astype(np.int32))
            assert int(pred[0]) == 0
            ys.append(y)
            ps.append(pred.cpu().numpy())
        y, p = np.concatenate(ys), np.concatenate(ps)
        conf = np.zeros((NL, NL), np.int64)
        np.add.at(conf, (y, p), 1)
        assert np.array_equal(met.conf.cpu().numpy(), conf)
        s = met.result()
        assert abs(s["acc"] - accuracy_score(y, p)) < 1e-12
        for avg in ("micro", "macro", "weighted"):
            assert abs(s[avg + "_f1"] - f1_score(y, p, average=avg)) < 1e-12
        _, pred_only = predict(dev(logits), want_probs=False)
        assert np.array_equal(pred_only.cpu().numpy(), ps[-1])


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 132, 1000), (37, 4, 64), (1030, 520, 333), (512, 1100, 256),
                                   (37, 1028, 64), (1030, 1540, 128)])
def test_gemm_bf16_nt_vs_fp64_on_rounded_operands(M, N, K):
    """Dense bf16 GEMM (LDS-DMA ring, XCD-aware tile map) vs an fp64 product of the SAME bf16-rounded operands: ragged
    M / N / K (rows beyond M and N are clamped loads, K is zero padded to 64), bias + LeakyReLU epilogue, and the
    adjacency helper built from cast_pad_bf16 + transpose_cast_bf16.  The last three shapes have fewer row blocks than XCDs
    and at least eight column tiles: the XCDs split the column tiles (the read-out of configs[4])."""
    rs = np.random.RandomState(M + N + K)
    a = rs.standard_normal((M, K)).astype(np.float32)
    b = rs.standard_normal((K, N)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    kp = (K + 63) // 64 * 64
    a_bf = ops.cast_pad_bf16(dev(a), ld=kp)
    bt_bf = ops.transpose_cast_bf16(dev(b))
    assert a_bf.shape == (M, kp) and bt_bf.shape == (N, kp)
    assert torch.equal(bt_bf[:, :K].float().cpu(), _bf16_round(b).t())
    assert float(bt_bf[:, K:].float().abs().max()) == 0.0 if kp > K else True
    ref = _bf16_round(a).double() @ _bf16_round(b).double()
    c = ops.gemm_bf16_nt(a_bf, bt_bf).cpu().double()
    assert float((c - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max())) * 1e-1
    c2 = ops.gemm_bf16_nt(a_bf, bt_bf, dev(bias), act=ops.ACT_LRELU2).cpu().double()
    ref2 = torch.nn.functional.leaky_relu(ref + torch.from_numpy(bias).double(), 0.2)
    assert float((c2 - ref2).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    if M == K:
        c3 = ops.dense_adj_matmul_bf16(a_bf, dev(b)).cpu().double()
        assert torch.equal(c3, c)


def test_bilstm_bf16_mfma_recurrence_vs_fp32_recurrence():
    """bf16-mode recurrence (W_hh . h of every step on v_mfma_f32_4x4x4_16b_bf16, bf16 operands, fp32 state) against the exact
    fp32 kernel on ragged batches: same zero padding, same bf16 side copy layout, bank within 5e-3 absolute (|h| < 1)."""
    import numpy as np
    rs = np.random.RandomState(5)
    for B, T in ((37, 100), (256, 100), (5, 24), (3, 230)):      # T = 230 > 200: the h rows leave LDS in two bursts
        V, E, Hh = 500, 300, 150
        lens = rs.randint(1, T + 1, size=B)
        lens[0], lens[-1] = T, 1
        if B == 5:
            lens[1] = 0                                            # an empty text: no rows, zero bank (both paths agree)
        tok = np.zeros((B, T), np.int64)
        for b in range(B):
            tok[b, :lens[b]] = rs.randint(1, V, size=lens[b])
        if B == 5:
            tok[2, 0], tok[3, 1] = V + 7, -3                        # out-of-range ids are clamped the same way by both paths
        emb = torch.from_numpy((0.4 * rs.standard_normal((V, E))).astype(np.float32)).to(DEV)
        weights = []
        for layer in range(2):
            for d in range(2):
                ind = E if layer == 0 else 2 * Hh
                weights.append(tuple(torch.from_numpy(rs.uniform(-0.08, 0.08, size=s).astype(np.float32)).to(DEV)
                                     for s in ((4 * Hh, ind), (4 * Hh, Hh), (4 * Hh,), (4 * Hh,))))
        t, l = torch.from_numpy(tok).to(DEV), torch.from_numpy(lens.astype(np.int64)).to(DEV)
        ref, ref_bf = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True)
        out, out_bf = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="bf16")
        torch.cuda.synchronize()
        assert out.shape == ref.shape and out_bf.shape == ref_bf.shape
        err = (out - ref).abs().max().item()
        assert err < 5e-3, err
        pad = torch.arange(T, device=DEV)[None, :] >= l[:, None]
        assert (out[pad] == 0).all() and (out_bf[pad] == 0).all() and (out_bf[:, :, 2 * Hh:] == 0).all()
        assert (out_bf[:, :, :2 * Hh].float() - out).abs().max().item() <= 2.0 ** -8       # the bf16 copy is the rounded bank
        if B == 37:                                            # single layer: the first layer is also the last (fp32 rows + bf16 copy)
            r1 = ops.bilstm(t, l, emb, weights[:2], Hh, 1)
            o1, o1b = ops.bilstm(t, l, emb, weights[:2], Hh, 1, want_bf16=True, recurrence="bf16")
            assert (o1 - r1).abs().max().item() < 5e-3 and (o1b[:, :, :2 * Hh].float() - o1).abs().max().item() <= 2.0 ** -8


def test_bilstm_bf16_layer0_projection_folded_into_the_embedding_table():
    """bf16 recurrence with the layer-0 input projection read out of table[v] = bf16(emb[v]) . bf16(W_ih0)^T + b_ih0 (folded once per
    weight version, mgnns_bilstm_bf16_fold_embedding) against the projection GEMM on the gathered rows in every forward: the same bank
    BIT FOR BIT (the GEMM adds an element's k terms in the same order whichever row it sits in), ragged / empty / clamped ids, one
    and two layers, long texts (more than 64 steps: the token ids travel 64 per register), and the table follows a weight update."""
    import numpy as np
    rs = np.random.RandomState(11)
    for B, T in ((37, 100), (256, 100), (5, 24), (3, 230)):
        V, E, Hh = 700, 300, 150
        lens = rs.randint(1, T + 1, size=B)
        lens[0], lens[-1] = T, 1
        if B == 5:
            lens[1] = 0
        tok = np.zeros((B, T), np.int64)
        for b in range(B):
            tok[b, :lens[b]] = rs.randint(1, V, size=lens[b])
        if B == 5:
            tok[2, 0], tok[3, 1] = V + 7, -3
        emb = torch.from_numpy((0.4 * rs.standard_normal((V, E))).astype(np.float32)).to(DEV)
        weights = []
        for layer in range(2):
            for d in range(2):
                ind = E if layer == 0 else 2 * Hh
                weights.append(tuple(torch.from_numpy(rs.uniform(-0.08, 0.08, size=s).astype(np.float32)).to(DEV)
                                     for s in ((4 * Hh, ind), (4 * Hh, Hh), (4 * Hh,), (4 * Hh,))))
        t, l = torch.from_numpy(tok).to(DEV), torch.from_numpy(lens.astype(np.int64)).to(DEV)
        for nl in (2, 1):
            w = weights[:2 * nl]
            cache = ops.LstmCache()
            ref, ref_bf = ops.bilstm(t, l, emb, w, Hh, nl, want_bf16=True, recurrence="bf16", cache=cache, fold=False)
            out, out_bf = ops.bilstm(t, l, emb, w, Hh, nl, want_bf16=True, recurrence="bf16", cache=cache, fold=True)
            assert cache.table is not None and tuple(cache.table[2].shape) == (V, 8 * Hh)
            assert torch.equal(out, ref) and torch.equal(out_bf, ref_bf), (B, T, nl)
            again, _ = ops.bilstm(t, l, emb, w, Hh, nl, want_bf16=True, recurrence="bf16", cache=cache, fold=True)
            assert torch.equal(again, out)
        if B == 37:                                                # a weight update: the table is rebuilt from the new version
            cache = ops.LstmCache()
            a0, _ = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="bf16", cache=cache, fold=True)
            first = cache.table[2]
            with torch.no_grad():
                emb.mul_(1.25)
            a1, _ = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="bf16", cache=cache, fold=True)
            b1, _ = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="bf16", cache=ops.LstmCache(), fold=False)
            assert cache.table[2] is not first and torch.equal(a1, b1) and not torch.equal(a1, a0)


def test_bilstm_prep_launch_builds_the_mask_packing_plan():
    """ops.bilstm(plan_mask=mask): the packing plan of the text mask built by an extra workgroup of the BiLSTM's prep launch is the
    stand-alone plan kernel's (ops.sq_mha_plan) int for int -- ragged lengths, a mask with holes and one that differs from the lengths
    (the plan is defined by the MASK), folded-table and per-forward projection forms -- and the bank is untouched by it."""
    import numpy as np
    rs = np.random.RandomState(31)
    V, E, Hh = 700, 300, 150
    emb = torch.from_numpy((0.4 * rs.standard_normal((V, E))).astype(np.float32)).to(DEV)
    weights = []
    for layer in range(2):
        for d in range(2):
            ind = E if layer == 0 else 2 * Hh
            weights.append(tuple(torch.from_numpy(rs.uniform(-0.08, 0.08, size=s).astype(np.float32)).to(DEV)
                                 for s in ((4 * Hh, ind), (4 * Hh, Hh), (4 * Hh,), (4 * Hh,))))
    for B, T in ((256, 100), (37, 128), (1, 9), (1000, 24)):
        lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, B))), 1, T).astype(int)
        lens[0] = T
        tok = np.zeros((B, T), np.int64)
        mask = np.zeros((B, T), np.float32)
        for b in range(B):
            tok[b, :lens[b]] = rs.randint(1, V, size=lens[b])
            mask[b, :lens[b]] = 1
        if B > 2:
            mask[1, ::2] = 0                                   # holes
            mask[2, :] = 0
            mask[2, min(T - 1, 5)] = 1                         # a mask that is not the length vector's
        t, l, m = torch.from_numpy(tok).to(DEV), torch.from_numpy(lens.astype(np.int64)).to(DEV), torch.from_numpy(mask).to(DEV)
        assert ops.bilstm_can_plan(B, T, E)
        want = ops.sq_mha_plan(m)
        for fold in (True, False):
            cache = ops.LstmCache()
            ref, ref_bf = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="bf16", cache=cache, fold=fold)
            out, out_bf, plan = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="bf16", cache=cache, fold=fold,
                                           plan_mask=m)
            # (group slots behind the last group are never written by either launch)
            ng = int(want[0])
            assert torch.equal(plan[:4 + 4 * ng], want[:4 + 4 * ng]) and torch.equal(plan[4 + 4 * B:], want[4 + 4 * B:]), (B, T, fold)
            assert torch.equal(out, ref) and torch.equal(out_bf, ref_bf)
    with pytest.raises(ValueError):
        ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="f32", plan_mask=m)


def test_bilstm_trailing_empty_samples_never_index_the_table_with_unwritten_tokens():
    """A padded partial batch (batching.BatchAssembler: trailing rows with lens == 0): the offset of an empty sample at the END of the
    batch is the total, a slot of pack_tok that prep never writes.  The workspace is poisoned with 0x7f bytes (the caching allocator
    hands the freed block back): the folded-table recurrence must not use that slot as a table row (0x7f7f7f7f * 4800 B is far
    outside the table), the bank of the empty samples is zero and the live samples equal the same samples run alone, bit for bit."""
    import numpy as np
    rs = np.random.RandomState(23)
    V, E, Hh = 700, 300, 150
    emb = torch.from_numpy((0.4 * rs.standard_normal((V, E))).astype(np.float32)).to(DEV)
    weights = []
    for layer in range(2):
        for d in range(2):
            ind = E if layer == 0 else 2 * Hh
            weights.append(tuple(torch.from_numpy(rs.uniform(-0.08, 0.08, size=s).astype(np.float32)).to(DEV)
                                 for s in ((4 * Hh, ind), (4 * Hh, Hh), (4 * Hh,), (4 * Hh,))))
    L = _lib.lib()
    for B, T, n_empty in ((8, 24, 3), (64, 100, 17), (2, 70, 1)):
        lens = rs.randint(1, T + 1, size=B)
        lens[0] = T
        lens[B - n_empty:] = 0
        tok = np.zeros((B, T), np.int64)
        for b in range(B):
            tok[b, :lens[b]] = rs.randint(1, V, size=lens[b])
        t, l = torch.from_numpy(tok).to(DEV), torch.from_numpy(lens.astype(np.int64)).to(DEV)
        live = B - n_empty
        for fold in (True, False):
            cache = ops.LstmCache()
            alone, alone_bf = ops.bilstm(t[:live].contiguous(), l[:live].contiguous(), emb, weights, Hh, 2, want_bf16=True,
                                         recurrence="bf16", cache=cache, fold=fold)
            torch.cuda.synchronize()
            for _ in range(3):
                poison = torch.full((max(L.mgnns_bilstm_workspace_bytes(B, T, Hh, 2), 16),), 0x7f, dtype=torch.uint8, device=DEV)
                torch.cuda.synchronize()
                del poison
                out, out_bf = ops.bilstm(t, l, emb, weights, Hh, 2, want_bf16=True, recurrence="bf16", cache=cache, fold=fold)
                torch.cuda.synchronize()
                assert (out[live:] == 0).all() and (out_bf[live:] == 0).all()
                assert torch.equal(out[:live], alone) and torch.equal(out_bf[:live], alone_bf), (B, T, fold)


@pytest.mark.parametrize("B,P,N", [(5, 196, 300), (2, 100, 300), (3, 224, 304), (1, 112, 17)])
def test_imgbank_pool_split_vs_fp64(B, P, N):
    """Split-bf16 image bank (csrc/imgbank_split.hip): bank within fp32-class error of fp64 (three bf16 MFMAs per product:
    ~2^-16 relative per product, K = 2048 products per output), per-half maxima exact."""
    rs = np.random.RandomState(P + N)
    K = 2048
    feat = np.maximum(rs.standard_normal((B, K, P)), 0).astype(np.float32)
    W = (0.05 * rs.standard_normal((N, K))).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    bank, pooled = ops.imgbank_pool_split(dev(feat), ops.pack_weight_bf16_split(dev(W)), dev(bias), N)
    ref = np.einsum("bkp,nk->bpn", feat.astype(np.float64), W.astype(np.float64)) + bias
    scale = np.einsum("bkp,nk->bpn", np.abs(feat).astype(np.float64), np.abs(W).astype(np.float64)).max()
    assert tuple(bank.shape) == (B, P, N)
    assert np.abs(bank.cpu().numpy() - ref).max() / scale < 2e-5
    assert torch.equal(pooled[:, 0], dev(feat[:, :, :112].max(2)))
    if P > 112:
        assert torch.equal(pooled[:, 1], dev(feat[:, :, 112:].max(2)))
    else:
        assert torch.isinf(pooled[:, 1]).all() and (pooled[:, 1] < 0).all()
    b2, _ = ops.imgbank_pool_split(dev(feat), ops.pack_weight_bf16_split(dev(W)), None, N, want_pool=False)
    assert np.abs(b2.cpu().numpy() - (ref - bias)).max() / scale < 2e-5
    if N % 2 == 0:
        # the same bank as split-bf16 images (hi = bf16(x), lo = bf16(x - hi), [2, B, P, 320], zero padded), with and without the
        # fp32 bank next to them: exactly the conversion pass's result on the fp32 bank (ops.split_pad_bf16)
        b3, p3, sp = ops.imgbank_pool_split(dev(feat), ops.pack_weight_bf16_split(dev(W)), dev(bias), N, want_split=True)
        assert torch.equal(b3, bank) and torch.equal(p3, pooled) and tuple(sp.shape) == (2, B, P, ops.BANK_LD)
        assert torch.equal(sp, ops.split_pad_bf16(bank))
        none, p4, sp4 = ops.imgbank_pool_split(dev(feat), ops.pack_weight_bf16_split(dev(W)), dev(bias), N, want_f32=False,
                                               want_split=True)
        assert none is None and torch.equal(sp4, sp) and torch.equal(p4, pooled)


@pytest.mark.parametrize("M,N,K", [(10000, 1024, 320), (5000, 512, 704), (2600, 128, 192), (512, 10000, 256), (300, 260, 64)])
def test_gemm_bf16_nt_last_round_k_split_equals_whole_tiles(M, N, K):
    """Dense bf16 GEMM with the workspace (a last, at most quarter-full round of tiles cut along K over the idle workgroups, partial
    sums added in part order by the fix-up launch) against the same GEMM without it: shapes whose XCDs end on a partial round
    (10 000 x 1024: 40 tiles on 32 workgroups per XCD), ragged row blocks, the column-split map (512 rows), bias + activation +
    bf16 output through the fix-up path; fp32 partial sums in another order only (<= 1e-6 of the output scale), repeatable bits."""
    rs = np.random.RandomState(M + N + K)
    a = ops.cast_pad_bf16(dev(rs.standard_normal((M, K)).astype(np.float32) * 0.1), ld=K)
    bt = ops.cast_pad_bf16(dev(rs.standard_normal((N, K)).astype(np.float32) * 0.1), ld=K)
    bias = dev(rs.standard_normal(N).astype(np.float32))
    old = ops.GEMM_BF16_KSPLIT
    try:
        for dt in (torch.float32, torch.bfloat16):
            ops.GEMM_BF16_KSPLIT = False
            ref = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            ops.GEMM_BF16_KSPLIT = True
            got = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            got2 = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            assert torch.equal(got, got2)
            scale = float(ref.float().abs().max())
            tol = 1e-6 if dt == torch.float32 else 8e-3          # (bf16 output: one ulp where a sum sits on a rounding boundary)
            assert float((got.float() - ref.float()).abs().max()) <= tol * scale, (dt, M, N, K)
    finally:
        ops.GEMM_BF16_KSPLIT = old


@pytest.mark.parametrize("form", [1, 3])
@pytest.mark.parametrize("M,N,K", [(10000, 1024, 2048), (2560, 256, 320), (2600, 1000, 704), (5000, 516, 1100), (2561, 260, 4200)])
def test_gemm_bf16_nt_160_and_320_tiles(M, N, K, form):
    """csrc/gemm_bf16.hip::gemm_bf16_nt_160_kernel (round 5: 160 x 256 tiles, producer waves, five-stage ring of 32-wide slices) and
    gemm_bf16_nt_320_kernel (320 x 256 tiles, sixteen waves, four-stage ring), whole tiles only, forced on (form 1 / 3), against round
    4's kernels (fp32 sums in another order only) and against fp64 on sampled entries: one and several tiles per workgroup, ragged
    M / N / K (rows beyond M clamped or read as zeros, a last row block of 1 row, N % 256 != 0, K padded to 64), bias + activation +
    bf16 output, repeatable bits."""
    rs = np.random.RandomState(M + N + K)
    kp = (K + 63) // 64 * 64
    a32 = rs.standard_normal((M, K)).astype(np.float32) * 0.1
    b32 = rs.standard_normal((N, K)).astype(np.float32) * 0.1
    a = ops.cast_pad_bf16(dev(a32), ld=kp)
    bt = ops.cast_pad_bf16(dev(b32), ld=kp)
    bias = dev(rs.standard_normal(N).astype(np.float32))
    try:
        for dt in (torch.float32, torch.bfloat16):
            ops.gemm_bf16_set_form(0)
            ref = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            ops.gemm_bf16_set_form(form)
            got = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            got2 = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            assert torch.equal(got, got2)
            if form in (1, 3):
                # round 6: the 160 x 256 / 320 x 256 tile with 64-wide K slices (whole 128-B operand lines; three / two stages; the default)
                # against the same tile with round 5's 32-wide slices: the same k-steps in the same order into the same accumulators -> the same bits
                ops.gemm_bf16_set_form(101)
                g32 = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
                ops.gemm_bf16_set_form(102)
                g64 = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
                ops.gemm_bf16_set_form(100)
                assert torch.equal(g32, g64) and torch.equal(g64, got), (dt, M, N, K)
            scale = float(ref.float().abs().max())
            tol = 2e-6 if dt == torch.float32 else 8e-3          # (bf16 output: one ulp where a sum sits on a rounding boundary)
            assert float((got.float() - ref.float()).abs().max()) <= tol * scale, (dt, M, N, K)
        plain = ops.gemm_bf16_nt(a, bt)                           # no bias, no activation: fp64 on sampled rows
        rows = rs.choice(M, size=min(M, 64), replace=False)
        rows[0], rows[1] = M - 1, 0
        want = _bf16_round(a32[rows]).double() @ _bf16_round(b32).double().t()
        assert float((plain[torch.from_numpy(rows).to(DEV)].cpu().double() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    finally:
        ops.gemm_bf16_set_form(100)
        ops.gemm_bf16_set_form(-1)


@pytest.mark.parametrize("M,N,K", [(10000, 2048, 1024), (10000, 1024, 320), (10000, 512, 2048), (1300, 260, 448)])
def test_gemm_bf16_nt_transposed_store(M, N, K):
    """gemm_bf16_nt(transposed_out=True) (round 6): the 160 x 256 kernel stores C^T [N, M] -- a product whose natural orientation has a small
    M runs as its transpose and still leaves the K-contiguous operand the next product needs.  Equal, bit for bit, to the transpose of
    the ordinary result of the same kernel and to the product computed in the OTHER orientation (operands swapped: the same dot
    products over k in the same order), fp32 and bf16, bias per column + activation, into a column slice of a wider matrix; shapes the
    kernel does not take are refused."""
    rs = np.random.RandomState(M + N + K)
    kp = (K + 63) // 64 * 64
    a = ops.cast_pad_bf16(dev(rs.standard_normal((M, K)).astype(np.float32) * 0.1), ld=kp)
    bt = ops.cast_pad_bf16(dev(rs.standard_normal((N, K)).astype(np.float32) * 0.1), ld=kp)
    bias = dev(rs.standard_normal(N).astype(np.float32))
    try:
        ops.gemm_bf16_set_form(1)
        for dt in (torch.float32, torch.bfloat16):
            ref = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            got = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt, transposed_out=True)
            assert tuple(got.shape) == (N, M) and torch.equal(got, ref.t().contiguous()), (dt, M, N, K)
            wide = torch.zeros(N, M + 64, device=DEV, dtype=dt)
            assert ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out=wide[:, :M], transposed_out=True).data_ptr() == wide.data_ptr()
            assert torch.equal(wide[:, :M], got) and float(wide[:, M:].abs().max()) == 0.0
        ops.gemm_bf16_set_form(-1)
        plain_t = ops.gemm_bf16_nt(a, bt, transposed_out=True)                       # [N, M] = (A . Bt^T)^T
        other = ops.gemm_bf16_nt(bt, a)                                              # [N, M] = Bt . A^T: the other orientation, any kernel
        assert float((plain_t - other).abs().max()) <= 2e-6 * float(other.abs().max())
    finally:
        ops.gemm_bf16_set_form(-1)
    with pytest.raises(ValueError):
        ops.gemm_bf16_nt(a[:640], bt, transposed_out=True)                          # four row blocks: not the 160 x 256 kernel's


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 4096), (10000, 2048, 4200), (8300, 2312, 4100)])
def test_gemm_bf16_nt_256_tiles(M, N, K):
    """Products with at least a full round of 256 x 256 tiles on every XCD and K >= 4096 run
    csrc/gemm_bf16.hip::gemm_bf16_nt_256_kernel (whole tiles round robin, the left-over tiles of an XCD cut along K and finished by
    the fix-up launch) when the workspace is there: 32 / 40 / 45 tiles per XCD = no remainder / 8 left-over tiles in 4 parts / 13
    in 2 parts; against the 256 x 128 kernel without workspace (fp32 sums in another order only), against fp64 on sampled entries,
    ragged M / N / K, bias + activation + bf16 output, repeatable bits."""
    rs = np.random.RandomState(M + N + K)
    kp = (K + 63) // 64 * 64
    a32 = rs.standard_normal((M, K)).astype(np.float32) * 0.1
    b32 = rs.standard_normal((N, K)).astype(np.float32) * 0.1
    a = ops.cast_pad_bf16(dev(a32), ld=kp)
    bt = ops.cast_pad_bf16(dev(b32), ld=kp)
    bias = dev(rs.standard_normal(N).astype(np.float32))
    old = ops.GEMM_BF16_KSPLIT
    try:
        for dt in (torch.float32, torch.bfloat16):
            ops.GEMM_BF16_KSPLIT = False
            ref = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            ops.GEMM_BF16_KSPLIT = True
            got = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            got2 = ops.gemm_bf16_nt(a, bt, bias, ops.ACT_LRELU2, out_dtype=dt)
            assert torch.equal(got, got2)
            scale = float(ref.float().abs().max())
            tol = 2e-6 if dt == torch.float32 else 8e-3
            assert float((got.float() - ref.float()).abs().max()) <= tol * scale, (dt, M, N, K)
        rows = rs.randint(0, M, size=48)
        rows[:4] = (0, M - 1, 255, 256)
        ref64 = _bf16_round(a32[rows]).double() @ _bf16_round(b32).double().t()
        got = ops.gemm_bf16_nt(a, bt).cpu().double()[rows]
        assert float((got - ref64).abs().max()) <= 1e-5 * max(1.0, float(ref64.abs().max()))
    finally:
        ops.GEMM_BF16_KSPLIT = old


def test_xcd_probe_confirms_the_block_index_placement():
    """`blockIdx.x & 7 == XCD` is an observed dispatch order several kernels place work by (SpMM feature slabs, the dense GEMM's
    row-block ranges): only speed depends on it, and mgnns_xcd_probe measures it on the device the tests run on -- eight distinct
    XCC_IDs, one per residue.  (A device where this fails still computes the same results; the assertion documents the box.)"""
    from mgnns_amd import _lib
    ok, ids = _lib.xcd_probe()
    assert len(ids) == 8
    assert ok and sorted(ids) == list(range(8)), ids
