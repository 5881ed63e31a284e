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
    def device_arrays(self, device):
        """(row_ptr, col, eid) int32 tensors on `device`, cached per device."""
        import torch
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(
                torch.from_numpy(a).to(device) for a in (self.row_ptr, self.col, self.eid))
        return self._dev[key]
