"""Seeded inputs of the per-operator golden cases -- TEST INFRASTRUCTURE.

Shared by oracle/gen_goldens.py (which feeds them to the reference classes) and by
tests/ (which regenerate them instead of storing megabytes of random inputs in the
fixtures).  numpy RandomState only.
"""
import numpy as np

MHA_B = 5
MHA_LENS = np.array([100, 1, 37, 4, 63])
MHA_CASES = [(H, tag, L, masked) for H in (1, 4, 8)
             for (tag, L, masked) in (("text", 100, True), ("img", 196, False))]


def mha_case(H, tag, L, masked):
    """-> q [B,300], bank [B,L,300], mask [B,L] or None (float32 ndarrays)."""
    rs = np.random.RandomState(9900 + 10 * H + (1 if masked else 0))
    q = (0.8 * rs.standard_normal((MHA_B, 300))).astype(np.float32)
    bank = (1.2 * rs.standard_normal((MHA_B, L, 300))).astype(np.float32)
    mask = None
    if masked:
        mask = np.zeros((MHA_B, L), dtype=np.float32)
        for b in range(MHA_B):
            mask[b, :MHA_LENS[b]] = 1.0
            bank[b, MHA_LENS[b]:] = 0.0      # padded LSTM rows are zero (MODEL:384)
    return q, bank, mask


def image_gcn_case(tag):
    """-> X [C,300], pooled [5,2048] for tag in {object, place}."""
    C, std, seed = {"object": (80, 0.45, 4301), "place": (365, 0.57, 4302)}[tag]
    rs = np.random.RandomState(seed)
    X = (std * rs.standard_normal((C, 300))).astype(np.float32)
    pooled = np.maximum(rs.standard_normal((5, 2048)), 0).astype(np.float32)
    return X, pooled


def gcn_projection():
    return np.random.RandomState(4242).standard_normal((2048, 8)).astype(np.float32)


def label_attention_key(tag):
    C, seed = {"object": (80, 7701), "place": (365, 7702)}[tag]
    return (3.0 * np.random.RandomState(seed).standard_normal((5, C))).astype(np.float32)
