"""Shared helpers for the parity tests (tests may use oracle/, the product may not)."""
import json
import os

import numpy as np
import torch

from mgnns_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def surface():
    with open(os.path.join(GOLDEN, "state_dict_surface.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def params_for(shapes, prefix="", skip=()):
    """{name: shape} -> {name: torch f32 tensor} via the by-name seeded fill."""
    return {prefix + k: torch.from_numpy(synth.param_value(prefix + k, s))
            for k, s in shapes.items() if k not in skip}


def mha_shapes(H, dk=128, D=300):
    return {
        "slf_attn.w_qs.weight": (H * dk, D), "slf_attn.w_qs.bias": (H * dk,),
        "slf_attn.w_ks.weight": (H * dk, D), "slf_attn.w_ks.bias": (H * dk,),
        "slf_attn.w_vs.weight": (H * dk, D), "slf_attn.w_vs.bias": (H * dk,),
        "slf_attn.layer_norm.gamma": (D,), "slf_attn.layer_norm.beta": (D,),
        "slf_attn.fc.weight": (D, H * dk), "slf_attn.fc.bias": (D,),
        "pos_ffn.w_1.weight": (D, D, 1), "pos_ffn.w_1.bias": (D,),
        "pos_ffn.w_2.weight": (D, D, 1), "pos_ffn.w_2.bias": (D,),
        "pos_ffn.layer_norm.gamma": (D,), "pos_ffn.layer_norm.beta": (D,),
    }


def label_attention_shapes(tag, C):
    s = {}
    for n, shp in (("w_q", (300, 300)), ("w_k", (300, C)), ("w_v", (300, C)), ("fc", (300, 300))):
        s["%s_attention.%s.weight" % (tag, n)] = shp
        s["%s_attention.%s.bias" % (tag, n)] = (300,)
    s["%s_linear_5.weight" % tag] = (100, 300)
    s["%s_linear_5.bias" % tag] = (100,)
    s["%s_x_linear.weight" % tag] = (300, 700)
    s["%s_x_linear.bias" % tag] = (300,)
    return s


def full_params(cfg, count, A_obj, A_place):
    """Parameters of the whole model by name, shapes derived from the committed
    state_dict surface (captured from the reference at cfg tumemo_b64: V=20154, H=4, NL=7)."""
    shapes = {}
    for k, s in surface().items():
        s = list(s)
        if k in ("embedding.weight", "text_features.node_hidden.weight"):
            s[0] = cfg.V
        elif k == "text_features.seq_edge_w.weight":
            s[0] = count
        elif k.endswith(("slf_attn.w_qs.weight", "slf_attn.w_ks.weight", "slf_attn.w_vs.weight",
                         "slf_attn.w_qs.bias", "slf_attn.w_ks.bias", "slf_attn.w_vs.bias")):
            s[0] = cfg.n_head * cfg.d_kv
        elif k.endswith("slf_attn.fc.weight"):
            s[1] = cfg.n_head * cfg.d_kv
        elif k in ("multi_linear_2.weight", "multi_linear_2.bias", "text_features.Linear.weight",
                   "text_features.Linear.bias"):
            s[0] = cfg.NL
        # stacks beyond stack_num do not exist
        parts = k.split(".")
        if parts[0].endswith("_multi_head_att") and len(parts) > 1 and parts[1].isdigit() \
                and int(parts[1]) >= cfg.stack_num:
            continue
        shapes[k] = tuple(s)
    p = params_for(shapes, skip=("object_A", "place_A"))
    p["object_A"] = torch.as_tensor(A_obj).float()
    p["place_A"] = torch.as_tensor(A_place).float()
    return p


def relerr(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def maxabs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())
