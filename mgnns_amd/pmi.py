"""Sparse PMI edge map (the `edges_matrix` argument of Text_GCN.Model).

The reference keeps the word-pair -> edge-weight-id map as a dense ``[V, V]`` int
matrix built by ``cal_PMI`` (reference utils/pmi.py:86-97, 3.25 GB at V=20154) and
indexes it one pair at a time from Python (reference models/Text_GCN.py:160-164).
Here the same map is a CSR structure (sorted columns per row) that lives in HBM
and is searched by the text-GCN kernel; id 0 means "no PMI entry" exactly as in
the dense matrix (utils/pmi.py:88 starts ``count`` at 1).

``PmiCsr`` also answers ``m[i, j]`` on the host, which is all the reference's
``Text_GCN.Model`` ever asks of its ``edges_matrix``.
"""
import numpy as np


class PmiCsr:
    def __init__(self, row_ptr, col, eid, n_rows):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.eid = np.ascontiguousarray(eid, dtype=np.int32)
        self.n_rows = int(n_rows)
        if self.row_ptr.shape[0] != self.n_rows + 1:
            raise ValueError("row_ptr must have n_rows+1 entries")
        if self.col.shape != self.eid.shape:
            raise ValueError("col and eid must have the same length")
        self._dev = {}

    # -- constructors -----------------------------------------------------
    @classmethod
    def from_dense(cls, m):
        """From the reference's dense ``edges_mappings`` (utils/pmi.py:86)."""
        m = np.asarray(m)
        if m.ndim != 2 or m.shape[0] != m.shape[1]:
            raise ValueError("edges_matrix must be square")
        rows, cols = np.nonzero(m)
        order = np.lexsort((cols, rows))
        rows, cols = rows[order], cols[order]
        row_ptr = np.zeros(m.shape[0] + 1, dtype=np.int64)
        np.add.at(row_ptr, rows + 1, 1)
        return cls(np.cumsum(row_ptr), cols, m[rows, cols], m.shape[0])

    @classmethod
    def from_coo(cls, rows, cols, eids, n_rows):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        eids = np.asarray(eids, dtype=np.int64)
        order = np.lexsort((cols, rows))
        rows, cols, eids = rows[order], cols[order], eids[order]
        if rows.size > 1 and np.any((rows[1:] == rows[:-1]) & (cols[1:] == cols[:-1])):
            raise ValueError("duplicate (row, col) pair in PMI map")
        row_ptr = np.zeros(n_rows + 1, dtype=np.int64)
        np.add.at(row_ptr, rows + 1, 1)
        return cls(np.cumsum(row_ptr), cols, eids, n_rows)

    @classmethod
    def coerce(cls, m):
        if isinstance(m, cls):
            return m
        if hasattr(m, "tocoo"):  # scipy sparse
            c = m.tocoo()
            return cls.from_coo(c.row, c.col, c.data, m.shape[0])
        return cls.from_dense(m)

    # -- host lookup (what the reference's Python loop calls) ---------------
    def __getitem__(self, ij):
        i, j = int(ij[0]), int(ij[1])
        lo, hi = int(self.row_ptr[i]), int(self.row_ptr[i + 1])
        k = lo + int(np.searchsorted(self.col[lo:hi], j))
        if k < hi and self.col[k] == j:
            return int(self.eid[k])
        return 0

    @property
    def shape(self):
        return (self.n_rows, self.n_rows)

    @property
    def nnz(self):
        return int(self.col.shape[0])

    def max_eid(self):
        return int(self.eid.max()) if self.eid.size else 0

    # -- device residency ---------------------------------------------------
    def eid_is_positional(self):
        """True when eid[k] == k + 1 for every stored entry -- the row-major numbering utils/pmi.py:86-97 hands out
        (and build_pmi reproduces).  The kernel then derives the id from the position and never reads `eid`."""
        return bool(np.array_equal(self.eid, np.arange(1, self.eid.shape[0] + 1, dtype=np.int32)))

    def device_arrays(self, device):
        """(row_ptr, col, eid) int32 tensors on `device`, cached per device; eid is None when ids are positional."""
        import torch
        key = str(device)
        if key not in self._dev:
            rp, col = (torch.from_numpy(a).to(device) for a in (self.row_ptr, self.col))
            if col.numel() == 0:
                col = torch.zeros(1, dtype=torch.int32, device=device)
            eid = None if self.eid_is_positional() else torch.from_numpy(self.eid).to(device)
            self._dev[key] = (rp, col, eid)
        return self._dev[key]


# ---------------------------------------------------------------------------------------------------------
# Building the edge map from a corpus (the reference's cal_PMI, utils/pmi.py:28-105) without the three dense
# [V,V] matrices and the four O(V^2) Python loops: pair counts are accumulated as sorted (src*V + dst) keys.
# Every quirk of the reference is kept, because edge ids index a learned embedding (seq_edge_w):
#   * sentences are padded with 'PAD' to 100 tokens, sentences of >= 100 tokens are dropped (pmi.py:7-15);
#   * the window is asymmetric: j in [i-window, i+window) (pmi.py:50-52);
#   * a PAD *target* is counted (it is a vocabulary word), an out-of-vocabulary source or target is skipped;
#     PAD never is a source and has word count 0, so its PMI is 0 (pmi.py:44-59,75-77);
#   * pairs seen fewer than min_cooccurence times are zeroed (pmi.py:60-66);
#   * PMI = log(p_ij / (p_i * p_j)) with counts normalised by the total word count, kept where > 0;
#   * ids are handed out in row-major (i, j) order starting at 1; id 0 = "no edge" (pmi.py:86-97).
# ---------------------------------------------------------------------------------------------------------
def build_pmi(texts, vocab, window_size=6, min_cooccurence=2, max_len=100, chunk=4096):
    """texts: iterable of whitespace-tokenised strings (the train split); vocab: list with 'PAD' at 0.
    Returns (edges_weights float32 [count, 1], PmiCsr, count) -- the reference's return triple with the
    dense edges_mappings replaced by its CSR."""
    V = len(vocab)
    d = {w: i for i, w in enumerate(vocab)}
    pad = d['PAD']
    rows = []
    for text in texts:
        words = text.split(' ')
        if len(words) >= max_len:
            continue
        ids = np.full(max_len, pad, dtype=np.int64)
        ids[:len(words)] = [d.get(w, -1) for w in words]
        # the reference compares the *string* with 'PAD': a literal "PAD" token in the text is padding too
        rows.append(ids)
    word_count = np.zeros(V, dtype=np.int64)
    keys_acc, cnts_acc = [], []
    offs = [o for o in range(-window_size, window_size) if o != 0]
    for c0 in range(0, len(rows), chunk):
        S = np.stack(rows[c0:c0 + chunk])                       # [n, max_len]
        src_ok = (S != pad) & (S >= 0)
        np.add.at(word_count, S[src_ok], 1)
        ks = []
        for o in offs:
            if o > 0:
                a, b, m = S[:, :-o], S[:, o:], src_ok[:, :-o]
            else:
                a, b, m = S[:, -o:], S[:, :o], src_ok[:, -o:]
            m = m & (b >= 0)
            ks.append(a[m] * V + b[m])
        k, c = np.unique(np.concatenate(ks), return_counts=True)
        keys_acc.append(k)
        cnts_acc.append(c)
    if keys_acc:
        allk = np.concatenate(keys_acc)
        allc = np.concatenate(cnts_acc)
        keys, inv = np.unique(allk, return_inverse=True)
        counts = np.zeros(keys.shape[0], dtype=np.int64)
        np.add.at(counts, inv, allc)
    else:
        keys = np.zeros(0, dtype=np.int64)
        counts = np.zeros(0, dtype=np.int64)
    keep = counts >= min_cooccurence
    keys, counts = keys[keep], counts[keep]
    total = np.sum(word_count)
    wc = word_count / total
    pij = counts / total
    i, j = keys // V, keys % V
    denom = wc[i] * wc[j]
    pmi = np.zeros(keys.shape[0], dtype=np.float64)
    ok = denom != 0
    pmi[ok] = np.log(pij[ok] / denom[ok])
    pmi = np.maximum(np.nan_to_num(pmi), 0.0)
    nz = pmi != 0
    i, j, pmi = i[nz], j[nz], pmi[nz]                            # keys are sorted -> row-major order
    eids = np.arange(1, pmi.shape[0] + 1)
    weights = np.concatenate([[0.0], pmi]).astype(np.float32).reshape(-1, 1)
    return weights, PmiCsr.from_coo(i, j, eids, V), int(pmi.shape[0] + 1)


def save_pmi(path, weights, pmi, count):
    """Compact on-disk form of (edges_weights, edges_mappings, count): one .npz, ~12 B per edge."""
    np.savez_compressed(path, row_ptr=pmi.row_ptr, col=pmi.col, eid=pmi.eid, weights=np.asarray(weights, np.float32),
                        n_rows=np.int64(pmi.n_rows), count=np.int64(count))


def load_pmi(path):
    z = np.load(path)
    return z["weights"], PmiCsr(z["row_ptr"], z["col"], z["eid"], int(z["n_rows"])), int(z["count"])
