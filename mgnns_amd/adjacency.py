"""Label-graph adjacency: gen_A (host, init time) and gen_adj (device, every forward).

Mirrors reference utils/util.py:382-398 (gen_A) and 421-426 (gen_adj).  gen_A runs once at
model construction on a [C,C] count matrix and stays on the host; gen_adj is part of the
forward (Multi_GCN_Multihead_att.py:461,490) and runs as a HIP kernel that also emits the
normalised adjacency in CSR for the sparse GraphConvolution step.
"""
import pickle

import numpy as np

from . import ops


def gen_A(num_classes, t, adj_file, gama=0.2):
    """Threshold / re-weight the co-occurrence counts -> (A [C,C] float64, nums [C,1]).

    Same arithmetic as utils/util.py:382-398.  The reference's call sites pass three
    arguments to a four-parameter function (Multi_GCN_Multihead_att.py:338,344); the missing
    `gama` is the paper's p = 0.2 (comment at util.py:396), which is the default here.
    """
    with open(adj_file, "rb") as f:
        result = pickle.load(f)
    adj = np.asarray(result["adj"], dtype=np.float64)
    nums = np.asarray(result["nums"], dtype=np.float64)[:, np.newaxis]
    if adj.shape != (num_classes, num_classes):
        raise ValueError("adjacency file %s holds %s, expected %d classes" % (adj_file, adj.shape, num_classes))
    adj = adj / nums
    adj = np.where(adj < t, 0.0, 1.0)
    adj = adj * gama / (adj.sum(0, keepdims=True) + 1e-6)
    adj = adj + (1 - gama) * np.identity(num_classes, np.int64)
    return adj, nums


def gen_adj(A):
    """D^-1/2 A^T D^-1/2 with D = diag(rowsum(A)) (utils/util.py:421-426), on the GPU."""
    return ops.gen_adj(A.detach().float().contiguous())


def gen_adj_csr(A):
    """(dense adj, (row_ptr, col, val)) -- the CSR feeds the sparse half of GraphConvolution."""
    return ops.gen_adj(A.detach().float().contiguous(), want_csr=True)
