"""One-off re-ordering of a static sparse adjacency for the LDS-tiled bf16 SpMM (csrc/spmm_bf16.hip,
`mgnns_spmm_tiled_bf16_fwd`): host-side integer plumbing, numpy only.

The label-graph adjacency is a model parameter (`gen_adj(A)`, utils/util.py:421-426; it changes when the weights do, not
per batch), so the CSR the reference would multiply by (`torch.matmul(adj, support)`, models/Multi_GCN_Multihead_att.py:54)
is re-ordered once into the stream the kernel walks:

  geometry   16 waves per workgroup, U rows per wave, tiles of BC columns, LB bytes of a row per lane
  record     all entries of (row block rb, wave w, column block cb), rows ascending, columns ascending inside a row
  part       <= 64 - 1 - HD entries of a record + a header: [n_entries | last << 31][HD = ceil(U / 4) dwords: one count byte
             per row of the wave][entries]; a record is one or more consecutive parts
  entry      (byte offset of the column's row inside the LDS tile) << 16 | bf16 value
  wave_off   [(rb * 16 + w) * (ncb + 1) + cb] -> dword offset of the record in `ent` (last slot of a row: end)

`reference_walk` replays the stream on the host exactly as the kernel does -- the CPU test of the format.
"""
import numpy as np

NW = 16          # waves per workgroup (csrc/spmm_bf16.hip)


def bf16_bits(x):
    """fp32 -> bf16 bit patterns (uint16), round to nearest even (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def bf16_to_f32(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def geometry_for(n_rows, F):
    """(lane_bytes, rows_per_wave, tile_cols) that fills the chip: one workgroup per CU holds R x FS accumulators, so
    R * FS ~ n_rows * F / 256."""
    lane_bytes, tile_cols = 8, 128                   # FS = 256 features, 512-B LDS rows, 64-KB tiles
    fs = 32 * lane_bytes
    if F % fs:
        raise ValueError("F=%d must be a multiple of %d for the tiled SpMM" % (F, fs))
    want_rows = max(1, -(-n_rows * (F // fs) // 256))          # rows per workgroup for ~256 workgroups
    rows_per_wave = 10 if want_rows <= 16 * 15 else 20
    return lane_bytes, rows_per_wave, tile_cols


class TiledPlan:
    def __init__(self, wave_off, ent, lane_bytes, rows_per_wave, tile_cols, n_rows, n_cols, nnz):
        self.wave_off, self.ent = wave_off, ent
        self.lane_bytes, self.rows_per_wave, self.tile_cols = lane_bytes, rows_per_wave, tile_cols
        self.n_rows, self.n_cols, self.nnz = n_rows, n_cols, nnz

    @property
    def n_row_blocks(self):
        return -(-self.n_rows // (NW * self.rows_per_wave))

    @property
    def n_col_blocks(self):
        return -(-self.n_cols // self.tile_cols)


def build_tiled_plan(row_ptr, col, val_bits, n_cols, lane_bytes, rows_per_wave, tile_cols):
    """CSR (row_ptr [n+1], col [nnz] ascending inside a row, val_bits [nnz] bf16 bit patterns) -> TiledPlan (numpy)."""
    rp = np.asarray(row_ptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    vb = np.asarray(val_bits, dtype=np.uint16)
    n = rp.size - 1
    U, BC = int(rows_per_wave), int(tile_cols)
    rowb = 64 * lane_bytes
    if BC * rowb > 65536:
        raise ValueError("tile of %d columns x %d B exceeds 64 KB" % (BC, rowb))
    HD = (U + 3) // 4
    EMAX = 64 - 1 - HD
    R = NW * U
    n_rb, ncb = -(-n // R), -(-n_cols // BC)
    nrec = n_rb * NW * ncb
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    if col.size and (col.min() < 0 or col.max() >= n_cols):
        raise ValueError("column index out of range")
    wv = rows // U                                       # global wave id = rb * 16 + w
    u = rows % U
    rec = wv * ncb + col // BC                           # record id
    order = np.lexsort((col, u, rec))                    # (record, row, column)
    rec_s, u_s = rec[order], u[order]
    entry = (((col[order] % BC) * rowb).astype(np.uint32) << np.uint32(16)) | vb[order].astype(np.uint32)
    tot = np.bincount(rec_s, minlength=nrec).astype(np.int64)            # entries per record
    nparts = np.maximum(1, -(-tot // EMAX))
    rec_len = tot + nparts * (1 + HD)
    rec_start = np.concatenate([[0], np.cumsum(rec_len)]).astype(np.int64)
    ent = np.zeros(int(rec_start[-1]) + 64, dtype=np.uint32)             # + 64: a part is always read as 64 dwords
    # position of every entry: record start + headers of the parts up to and including its own + index inside the record
    first = np.concatenate([[0], np.cumsum(tot)])[:-1]                    # index of a record's first entry in sorted order
    idx_in_rec = np.arange(rec_s.size, dtype=np.int64) - first[rec_s]
    part = idx_in_rec // EMAX
    ent[rec_start[rec_s] + (part + 1) * (1 + HD) + idx_in_rec] = entry
    # part headers
    part_first = np.concatenate([[0], np.cumsum(nparts)])                 # global part id of a record's first part
    npart_tot = int(part_first[-1])
    part_rec = np.repeat(np.arange(nrec, dtype=np.int64), nparts)
    part_k = np.arange(npart_tot, dtype=np.int64) - part_first[part_rec]
    part_n = np.minimum(EMAX, tot[part_rec] - part_k * EMAX).clip(min=0)
    part_pos = rec_start[part_rec] + part_k * (1 + HD) + part_k * EMAX
    last = (part_k == nparts[part_rec] - 1).astype(np.uint32)
    ent[part_pos] = part_n.astype(np.uint32) | (last << np.uint32(31))
    gpart = part_first[rec_s] + part                                      # global part id of every entry
    cnt = np.bincount(gpart * U + u_s, minlength=npart_tot * U).reshape(npart_tot, U).astype(np.uint32)
    if cnt.size and cnt.max() > 255:
        raise AssertionError("count byte overflow")                       # impossible: EMAX < 64
    cb4 = np.zeros((npart_tot, HD * 4), dtype=np.uint32)
    cb4[:, :U] = cnt
    hdr = (cb4[:, 0::4] | (cb4[:, 1::4] << np.uint32(8)) | (cb4[:, 2::4] << np.uint32(16)) | (cb4[:, 3::4] << np.uint32(24)))
    for i in range(HD):
        ent[part_pos + 1 + i] = hdr[:, i]
    wave_off = np.zeros((n_rb * NW, ncb + 1), dtype=np.uint32)
    wave_off[:, :ncb] = rec_start[:-1].reshape(n_rb * NW, ncb)
    wave_off[:, ncb] = rec_start[1:].reshape(n_rb * NW, ncb)[:, -1]
    if rec_start[-1] + 64 >= 2 ** 32:
        raise ValueError("plan too large for 32-bit offsets")
    return TiledPlan(wave_off.reshape(-1), ent, lane_bytes, U, BC, n, n_cols, int(col.size))


def reference_walk(plan, X_bits, F, act=0):
    """Replay the plan like spmm_bf16_tiled_kernel (same traversal, fp32 accumulation in stream order) -> Y fp32 [n, F].
    Test infrastructure for the FORMAT (slow: python loops over records)."""
    U, BC, rowb = plan.rows_per_wave, plan.tile_cols, 64 * plan.lane_bytes
    HD = (U + 3) // 4
    ncb = plan.n_col_blocks
    X = bf16_to_f32(np.asarray(X_bits, dtype=np.uint16).reshape(plan.n_cols, F))
    Y = np.zeros((plan.n_rows, F), dtype=np.float32)
    ent, wo = plan.ent, plan.wave_off.reshape(-1, ncb + 1)
    for wv in range(wo.shape[0]):
        for cb in range(ncb):
            pos = int(wo[wv, cb])
            while True:
                hdr = int(ent[pos])
                counts = ent[pos + 1:pos + 1 + HD].view(np.uint8)[:U]
                j = pos + 1 + HD
                for uu in range(U):
                    for _ in range(int(counts[uu])):
                        e = int(ent[j])
                        j += 1
                        c = cb * BC + (e >> 16) // rowb
                        w = bf16_to_f32(np.uint16(e & 0xFFFF))
                        r = wv * U + uu
                        Y[r] = Y[r] + (w * X[c]).astype(np.float32)     # bf16 x bf16 products are exact in fp32
                assert j - (pos + 1 + HD) == (hdr & 0x7FFFFFFF)
                if hdr >> 31:
                    break
                pos = j
            assert pos <= int(wo[wv, cb + 1]) if cb + 1 <= ncb else True
    if act == 1:
        Y = np.maximum(Y, 0)
    elif act == 2:
        Y = np.where(Y > 0, Y, np.float32(0.2) * Y)
    return Y
