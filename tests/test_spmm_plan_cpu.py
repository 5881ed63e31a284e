"""The re-ordered adjacency stream of the LDS-tiled bf16 SpMM (mgnns_amd/spmm_plan.py) on the CPU: `reference_walk` replays
the stream exactly as csrc/spmm_bf16.hip::spmm_bf16_tiled_kernel walks it; the result must be the sparse product
(GraphConvolution.forward's `torch.matmul(adj, support)`, models/Multi_GCN_Multihead_att.py:54) on the bf16-rounded operands."""
import numpy as np
import pytest

from mgnns_amd import spmm_plan as sp
from mgnns_amd.stress import random_csr


def _ref(rp, col, vbits, xbits, n, ncols, F, act):
    X = sp.bf16_to_f32(xbits).reshape(ncols, F).astype(np.float64)
    v = sp.bf16_to_f32(vbits).astype(np.float64)
    out = np.zeros((n, F))
    for r in range(n):
        lo, hi = rp[r], rp[r + 1]
        out[r] = (v[lo:hi, None] * X[col[lo:hi]]).sum(0)
    if act == 2:
        out = np.where(out > 0, out, 0.2 * out)
    return out


@pytest.mark.parametrize("n,dens,geo,F", [(700, 0.05, (8, 10, 128), 256), (333, 0.5, (8, 20, 128), 256),
                                           (500, 0.02, (4, 20, 256), 128), (161, 0.2, (4, 10, 256), 128)])
def test_plan_replay_equals_the_sparse_product(n, dens, geo, F):
    rs = np.random.RandomState(n)
    rp, col, val = random_csr(n, dens, n)
    vb = sp.bf16_bits(val)
    xb = sp.bf16_bits(rs.randn(n, F).astype(np.float32))
    plan = sp.build_tiled_plan(rp, col, vb, n, *geo)
    assert plan.nnz == col.size and plan.ent.size >= col.size + 64
    if dens >= 0.2:                                  # several parts per record: the multi-part path is exercised
        hd = (geo[1] + 3) // 4
        assert (plan.ent[plan.wave_off.reshape(-1, plan.n_col_blocks + 1)[:, :-1].ravel()] >> 31 == 0).any() or 64 - 1 - hd >= geo[1] * geo[2]
    y = sp.reference_walk(plan, xb, F, act=2)
    ref = _ref(rp, col, vb, xb, n, n, F, 2)
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()


def test_plan_handles_empty_rows_rectangular_and_ragged_shapes():
    rs = np.random.RandomState(5)
    n, ncols, F = 205, 390, 256                      # not multiples of the row block (160) / the tile (128)
    dense = (rs.rand(n, ncols) < 0.08) * rs.rand(n, ncols)
    dense[7] = 0
    dense[n - 1] = 0                                 # empty rows, the last one included
    dense[11, :] = rs.rand(ncols)                    # a full row: 390 entries of one wave in 4 column blocks
    rp = np.concatenate([[0], np.cumsum((dense != 0).sum(1))]).astype(np.int32)
    col = np.nonzero(dense)[1].astype(np.int32)
    val = dense[dense != 0].astype(np.float32)
    vb, xb = sp.bf16_bits(val), sp.bf16_bits(rs.randn(ncols, F).astype(np.float32))
    plan = sp.build_tiled_plan(rp, col, vb, ncols, 8, 10, 128)
    assert plan.n_row_blocks == 2 and plan.n_col_blocks == 4
    y = sp.reference_walk(plan, xb, F)
    ref = _ref(rp, col, vb, xb, n, ncols, F, 0)
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()
    assert not y[7].any() and not y[n - 1].any()


def test_bf16_bits_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.140625, 1e-40, 65504.0], dtype=np.float32)
    b = sp.bf16_bits(x)
    import torch
    t = torch.from_numpy(x).bfloat16().view(torch.int16).numpy().view(np.uint16)
    assert (b == t).all()
    assert (sp.bf16_to_f32(b) == torch.from_numpy(x).bfloat16().float().numpy()).all()


def test_geometry_fills_the_chip():
    assert sp.geometry_for(10000, 1024) == (8, 10, 128)     # 63 row blocks x 4 slabs = 252 workgroups
    assert sp.geometry_for(10000, 2048) == (8, 20, 128)     # 32 x 8 = 256
    with pytest.raises(ValueError):
        sp.geometry_for(1000, 300)
    with pytest.raises(ValueError):
        sp.build_tiled_plan(np.array([0, 1]), np.array([5]), np.array([1], np.uint16), 4, 8, 10, 128)   # column out of range
