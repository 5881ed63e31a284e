"""Synthetic-workload harness shared by bench.py, __graft_entry__.smoke() and the GPU tests:
builds the product module with by-name seeded parameters and the 7-argument call of
engine/Multi_GCN_Multihead_Att_engine.py:825."""
import torch

from . import synth
from .model import Multi_GCN_Multihead_Att, Text_model_from_parts


def make_vocab(V):
    return ["PAD", "UNK"] + ["w%d" % i for i in range(2, V)]


def build_model(cfg, pmi, count, A_obj, A_place, label_query, device=None, trunks=False):
    """trunks=True also builds the reference's CNN trunks (MODEL:629-630: ResNet-101 for objects, ResNet-50 with 365
    classes for places) with seeded weights, so the model accepts raw [B,3,448,448] images."""
    tm = Text_model_from_parts(make_vocab(cfg.V), pmi, count, cfg.NL, cfg.ngram, 0.5)
    obj = place = None
    if trunks:
        from . import trunk
        obj, place = trunk.resnet101(), trunk.resnet50(num_classes=365)
    m = Multi_GCN_Multihead_Att(cfg.opt(), cfg.NL, tm, obj, place, cfg.C_obj, cfg.C_place,
                                label_glove=torch.as_tensor(label_query))
    sd = {}
    for k, v in m.state_dict().items():
        if k == "object_A":
            sd[k] = torch.as_tensor(A_obj).float()
        elif k == "place_A":
            sd[k] = torch.as_tensor(A_place).float()
        elif k.startswith("object_features.") or k.startswith("place_features."):
            sd[k] = torch.from_numpy(synth.trunk_param_value(k, tuple(v.shape)))
        else:
            sd[k] = torch.from_numpy(synth.param_value(k, tuple(v.shape)))
    m.load_state_dict(sd, strict=True)
    m.eval()
    if device is not None:
        m = m.to(device)
    return m


def synthetic_adjacencies(cfg, seed=5):
    """Object / place adjacency with the sparsity of the shipped data (53 / 217 off-diagonal edges)."""
    return (synth.synth_adjacency(cfg.C_obj, 53 if cfg.C_obj == 80 else max(1, cfg.C_obj // 2), seed),
            synth.synth_adjacency(cfg.C_place, 217 if cfg.C_place == 365 else max(1, cfg.C_place // 2), seed + 1))


def call_args(inp, device):
    """The 7 positional arguments of the engine's model(...) call, on `device`."""
    t = {k: torch.as_tensor(v).to(device) for k, v in inp.items()}
    return (t["text"], t["text_lens"], t["text_mask"], t["object_feature"], t["place_feature"],
            t["object_inp"], t["place_inp"])
