"""Seeded synthetic data in the shapes of the reference's datasets.

Everything is drawn from ``numpy.random.RandomState`` (legacy MT19937, stable across
numpy versions) so that golden fixtures only have to store seeds and outputs:

* parameters: one stream per parameter NAME (seed = crc32(name)), see ``param_value``;
* inputs: one stream per (config, seed).

Shapes follow the reference's data path: token ids / lens / mask as produced by
utils/Multi_GCN_Co_att_dataset.py:233-265, feature maps as produced by the ResNet
trunks (models/Multi_GCN_Multihead_att.py:450,482), label/object/place GloVe inputs
as the pickles under data/, adjacency through the gen_A formula (utils/util.py:382-398),
PMI edge map with the id convention of utils/pmi.py:86-97.
"""
import zlib
from dataclasses import dataclass, field

import numpy as np

from .pmi import PmiCsr


# ---------------------------------------------------------------------------
# parameters by name
# ---------------------------------------------------------------------------
def _rs(name, salt=0):
    return np.random.RandomState((zlib.crc32(name.encode()) + salt) & 0x7FFFFFFF)


def param_value(name, shape, salt=0):
    """Deterministic value for the parameter called `name` (a state_dict key)."""
    shape = tuple(int(s) for s in shape)
    rs = _rs(name, salt)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "gamma":                       # custom LayerNorm scale (submodules.py:148)
        return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
    if leaf == "beta":
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    if name.endswith("seq_edge_w.weight"):    # learnable PMI edge weights (Text_GCN.py:68)
        return rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if name.endswith("node_hidden.weight") or name == "embedding.weight":
        w = (0.4 * rs.standard_normal(shape)).astype(np.float32)
        if name == "embedding.weight":
            w[0] = 0.0                        # padding_idx row (Multi_GCN_Multihead_att.py:364)
        return w
    if name.startswith("lstm.") or name.startswith("rnn."):
        return rs.uniform(-0.08, 0.08, size=shape).astype(np.float32)
    if "bias" in leaf:
        return (0.05 * rs.standard_normal(shape)).astype(np.float32)
    return (0.05 * rs.standard_normal(shape)).astype(np.float32)


def fill_state_dict(shapes, skip=(), salt=0):
    """{name: shape} -> {name: float32 ndarray}; names in `skip` are left out."""
    return {k: param_value(k, s, salt) for k, s in shapes.items() if k not in skip}


def trunk_param_value(name, shape, salt=0):
    """Seeded values for a ResNet trunk state_dict entry (torchvision key names): He-scaled convolution weights and
    BatchNorm statistics that keep activations O(1) through 33 residual blocks (the last BatchNorm of a block and
    of a projection get a small scale, as trained networks have)."""
    shape = tuple(int(s) for s in shape)
    rs = _rs(name, salt)
    leaf = name.rsplit(".", 1)[-1]
    owner = name.rsplit(".", 2)[-2] if name.count(".") else ""
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        return (np.sqrt(2.0 / fan_in) * rs.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if leaf == "running_mean":
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:
        small = owner == "bn3"
        return rs.uniform(0.2, 0.4, size=shape).astype(np.float32) if small else rs.uniform(0.6, 1.0, size=shape).astype(np.float32)
    if leaf == "bias" and len(shape) == 1:
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    return (0.02 * rs.standard_normal(shape)).astype(np.float32)


def fill_trunk_(module, salt=0):
    """Load trunk_param_value() into every entry of module.state_dict() (in place); returns the module."""
    import torch
    sd = module.state_dict()
    module.load_state_dict({k: torch.from_numpy(trunk_param_value(k, v.shape, salt)) for k, v in sd.items()})
    return module


# ---------------------------------------------------------------------------
# adjacency (gen_A formula) and PMI map
# ---------------------------------------------------------------------------
def gen_A_from_binary(P, gamma=0.2):
    """A = P*gamma/(colsum(P)+1e-6) + (1-gamma)*I   (utils/util.py:395-397)."""
    P = np.asarray(P, dtype=np.float64)
    A = P * gamma / (P.sum(0, keepdims=True) + 1e-6)
    return (A + (1.0 - gamma) * np.identity(P.shape[0])).astype(np.float32)


def synth_adjacency(C, nnz_offdiag, seed, gamma=0.2):
    """Random binary co-occurrence with `nnz_offdiag` off-diagonal ones -> A [C,C] f32."""
    rs = np.random.RandomState(seed)
    P = np.zeros((C, C))
    n = 0
    while n < nnz_offdiag:
        i, j = rs.randint(0, C, size=2)
        if i != j and P[i, j] == 0:
            P[i, j] = 1.0
            n += 1
    return gen_A_from_binary(P, gamma)


def synth_pmi(V, per_row=8, seed=7, diag_frac=0.05):
    """Random PMI edge map: ~per_row entries per row, ids 1..count-1 in row-major
    order of appearance (utils/pmi.py:89-96), a few diagonal entries.  Returns
    (PmiCsr, count)."""
    rs = np.random.RandomState(seed)
    n = rs.poisson(per_row, size=V)
    n[:2] = 0                                   # PAD / UNK rows carry no PMI
    rows = np.repeat(np.arange(V), n)
    cols = rs.randint(2, V, size=rows.size)
    diag = np.nonzero(rs.uniform(size=V) < diag_frac)[0]
    diag = diag[diag >= 2]
    rows = np.concatenate([rows, diag])
    cols = np.concatenate([cols, diag])
    key = np.unique(rows.astype(np.int64) * V + cols)   # sorted = row-major order
    rows, cols = key // V, key % V
    eids = np.arange(1, key.size + 1)
    return PmiCsr.from_coo(rows, cols, eids, V), int(key.size + 1)


def synth_tokens(B, T, V, pmi, seed, walk=0.6, repeat=0.08, force_extremes=True):
    """Token ids [B,T] int64 (0 = PAD, right padded), lens [B] int64, mask [B,T] f32.

    Lengths follow clip(round(exp(N(2.4,0.75))),4,T) (TumEmo: mean 16.1, max 100).
    Tokens are a random walk over the PMI graph (so edge lookups actually hit),
    mixed with uniform draws and repeats of earlier tokens.
    """
    rs = np.random.RandomState(seed)
    lens = np.clip(np.round(np.exp(rs.normal(2.4, 0.75, size=B))), 4, T).astype(np.int64)
    if force_extremes and B >= 2:
        lens[0] = T
        lens[1] = min(4, T)
    tok = np.zeros((B, T), dtype=np.int64)
    rp, col = pmi.row_ptr, pmi.col
    for b in range(B):
        prev = []
        for i in range(int(lens[b])):
            u = rs.uniform()
            t = 0
            if prev and u < repeat:
                t = prev[rs.randint(len(prev))]
            elif prev and u < repeat + walk:
                src = prev[-1 - rs.randint(min(len(prev), 3))]
                lo, hi = rp[src], rp[src + 1]
                if hi > lo:
                    t = int(col[lo + rs.randint(hi - lo)])
            if t == 0:
                t = int(rs.randint(2, V))
            tok[b, i] = t
            prev.append(t)
    mask = (tok != 0).astype(np.float32)
    return tok, lens, mask


# ---------------------------------------------------------------------------
# configurations
# ---------------------------------------------------------------------------
@dataclass
class Config:
    name: str
    B: int
    T: int = 100
    V: int = 20154
    NL: int = 7            # num_labels: width of the logits
    NLQ: int = 7           # rows of the label-GloVe query (hard-coded 7 in the reference, MODEL:101)
    n_head: int = 4
    stack_num: int = 2
    d_kv: int = 128
    ngram: int = 4
    C_obj: int = 80
    C_place: int = 365
    hidden_size: int = 150
    emb_size: int = 300
    num_layers: int = 2
    seed: int = 1234
    extra: dict = field(default_factory=dict)

    def opt(self):
        """The `opt` dict of Tumblr_Multi_GCN_Multihead_Att.py:100-112."""
        return {"emb_path": None, "bidirectional": True, "hidden_size": self.hidden_size,
                "emb_size": self.emb_size, "num_layers": self.num_layers, "dropout": 0.5,
                "emb_type": "random", "vocab_size": self.V, "stack_num": self.stack_num,
                "n_head": self.n_head, "d_kv": self.d_kv, "is_regu": False}


CONFIGS = {
    # cfg 1: plumbing, CPU-sized
    "mvsa_single_b8": Config("mvsa_single_b8", B=8, T=50, V=6000, NL=3, n_head=1, stack_num=1, seed=1235),
    # cfg 2: TumEmo-shaped
    "tumemo_b64": Config("tumemo_b64", B=64, T=100, V=20154, NL=7, n_head=4, stack_num=2, seed=1236),
    # cfg 3: MVSA-Multiple-shaped (the headline)
    "mvsa_multiple_b256": Config("mvsa_multiple_b256", B=256, T=100, V=20154, NL=3, n_head=8,
                                 stack_num=2, seed=1237),
}


def make_inputs(cfg, B=None, seed=None, pmi=None, feature_hw=14):
    """Synthetic forward inputs for `cfg` as numpy arrays (dict).

    Keys follow the 7-argument call of engine/Multi_GCN_Multihead_Att_engine.py:825:
    text, text_lens, text_mask, object_feature, place_feature, object_inp, place_inp
    (feature maps are the post-trunk [B,2048,14,14] tensors), plus label_query.
    """
    B = cfg.B if B is None else B
    seed = cfg.seed if seed is None else seed
    rs = np.random.RandomState(seed)
    if pmi is None:
        pmi, _ = synth_pmi(cfg.V, seed=seed + 17)
    tok, lens, mask = synth_tokens(B, cfg.T, cfg.V, pmi, seed + 1)
    P = feature_hw * feature_hw
    obj = np.maximum(rs.standard_normal((B, 2048, P)).astype(np.float32), 0.0)
    plc = np.maximum(rs.standard_normal((B, 2048, P)).astype(np.float32), 0.0)
    obj_inp = (0.45 * rs.standard_normal((cfg.C_obj, 300))).astype(np.float32)
    plc_inp = (0.57 * rs.standard_normal((cfg.C_place, 300))).astype(np.float32)
    label = (0.35 * rs.standard_normal((cfg.NLQ, 300))).astype(np.float32)
    return {
        "text": tok, "text_lens": lens, "text_mask": mask,
        "object_feature": obj.reshape(B, 2048, feature_hw, feature_hw),
        "place_feature": plc.reshape(B, 2048, feature_hw, feature_hw),
        "object_inp": np.broadcast_to(obj_inp, (B,) + obj_inp.shape).copy(),
        "place_inp": np.broadcast_to(plc_inp, (B,) + plc_inp.shape).copy(),
        "label_query": label,
    }
