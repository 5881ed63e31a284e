"""CPU restatement of the MGNNS forward hot path -- TEST INFRASTRUCTURE ONLY.

This file is the parity oracle: an independent torch-CPU / numpy restatement of the
reference's forward arithmetic.  It is imported only by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and only as
the checker / the timed CPU baseline.  The product (mgnns_amd/) never imports it and
has no CPU path of its own.

Pinning: every function below is checked in tests/test_oracle_golden.py against
golden vectors produced by the *unmodified reference classes* imported in the build
container (oracle/gen_goldens.py; fixtures under tests/golden/).  The text-GCN
channel depends on DGL, which is not vendored in the reference and not installed
here: its goldens were produced with the documented DGL semantics stand-in in
oracle/ref_shims.py (message = h_src * w_edge, reduce = max, zero in-degree -> 0,
sum_nodes = per-graph sum), so that one channel is pinned to the reference's own
Python graph construction but "parity unpinned" with respect to DGL's kernels.

All functions are functional: parameters come in a dict keyed by the reference's
state_dict names holding float32 torch CPU tensors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------
# A.7 / A.3  adjacency
# ---------------------------------------------------------------------------
def gen_A(adj_counts, nums, t, gamma=0.2):
    """Reference utils/util.py:382-398 (gen_A).  counts [C,C], nums [C] -> A [C,C] f64."""
    adj = np.asarray(adj_counts, dtype=np.float64) / np.asarray(nums, dtype=np.float64)[:, None]
    adj = np.where(adj < t, 0.0, 1.0)
    adj = adj * gamma / (adj.sum(0, keepdims=True) + 1e-6)
    return adj + (1.0 - gamma) * np.identity(adj.shape[0], dtype=np.int64)


def gen_adj(A):
    """Reference utils/util.py:421-426: D = diag(rowsum(A)^-1/2); adj = (A D)^T D."""
    D = torch.pow(A.sum(1).float(), -0.5)
    D = torch.diag(D)
    return torch.matmul(torch.matmul(A, D).t(), D)


def graph_convolution(x, adj, weight):
    """Reference models/Multi_GCN_Multihead_att.py:52-58: support = X W, then adj @ support."""
    return torch.matmul(adj, torch.matmul(x, weight))


def image_gcn(A, inp, gc1_w, gc2_w):
    """MODEL:461-473 / 490-499: G = gc2(LeakyReLU_0.2(gc1(X, adj)), adj)  -> [C, 2048]."""
    adj = gen_adj(A)
    x = graph_convolution(inp, adj, gc1_w)
    x = F.leaky_relu(x, 0.2)
    return graph_convolution(x, adj, gc2_w)


# ---------------------------------------------------------------------------
# A.3  image memory bank + max-pool read-out
# ---------------------------------------------------------------------------
def img_memory_bank(feat, weight, bias):
    """MODEL:400-428: feat [B,2048,P] -> bank[b,p,:] = W feat[b,:,p] + c  -> [B,P,300]."""
    B = feat.shape[0]
    x = feat.reshape(B, 2048, -1).permute(0, 2, 1).reshape(-1, 2048)
    return F.linear(x, weight, bias).view(B, -1, weight.shape[0])


def max_pool(feat):
    """MODEL:302,454-455: MaxPool2d(14,14) on [B,2048,14,14] -> [B,2048]."""
    B = feat.shape[0]
    return feat.reshape(B, 2048, -1).max(dim=2).values


# ---------------------------------------------------------------------------
# A.4  label-query "attention"
# ---------------------------------------------------------------------------
def label_attention(p, prefix, query, key, n_heads=5, mask=None):
    """MODEL:88-133.  query [NLQ,300] (any float dtype), key=value [B,C] -> [B,NLQ,300].
    mask (MODEL:118-119; no call site of the reference passes one, so no golden vector pins this branch):
    broadcast against the energy [B,NLQ,heads,dh], zeros are filled with -1e10 in front of the softmax."""
    hid = p[prefix + ".w_q.weight"].shape[0]
    dh = hid // n_heads
    Q = F.linear(query.float(), p[prefix + ".w_q.weight"], p[prefix + ".w_q.bias"])
    K = F.linear(key, p[prefix + ".w_k.weight"], p[prefix + ".w_k.bias"])
    V = F.linear(key, p[prefix + ".w_v.weight"], p[prefix + ".w_v.bias"])
    NLQ, B = Q.shape[0], K.shape[0]
    Q = Q.view(1, NLQ, n_heads, dh)
    K = K.view(B, 1, n_heads, dh)
    V = V.view(B, 1, n_heads, dh)
    energy = (Q * K) / torch.sqrt(torch.tensor([float(dh)]))
    if mask is not None:
        energy = energy.masked_fill(mask == 0, -1e10)
    att = torch.softmax(energy, dim=-1)
    x = (att * V).contiguous().view(B, NLQ, hid)
    return F.linear(x, p[prefix + ".fc.weight"], p[prefix + ".fc.bias"])


def label_attention_tail(p, chan, y):
    """MODEL:477-479 / 504-506: linear_5 -> view [B, 100*NLQ] -> x_linear."""
    B = y.shape[0]
    z = F.linear(y, p[chan + "_linear_5.weight"], p[chan + "_linear_5.bias"]).reshape(B, -1)
    return F.linear(z, p[chan + "_x_linear.weight"], p[chan + "_x_linear.bias"])


# ---------------------------------------------------------------------------
# A.5  single-query multi-head attention layer
# ---------------------------------------------------------------------------
def layer_norm(x, gamma, beta, eps=1e-6):
    """Reference models/submodules.py:153-156: unbiased std, eps added to std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return gamma * (x - mean) / (std + eps) + beta


def sq_mha_layer(p, prefix, q, bank, mask, n_head, d_kv, return_head_diff=False):
    """MyMultiHeadAttention.forward (moudles.py:207-230) with len_q == 1.

    q [B,300], bank [B,L,300] (key = value), mask [B,L] float (1 token / 0 pad) or None.
    Returns (out [B,300], attn [H*B,1,L]) with attn rows in head-major order h*B+b
    (submodules.py:72-78).
    """
    a = prefix + ".slf_attn."
    f = prefix + ".pos_ffn."
    B, L, D = bank.shape
    H, dk = n_head, d_kv
    qh = F.linear(q, p[a + "w_qs.weight"], p[a + "w_qs.bias"]).view(B, H, dk)
    kh = F.linear(bank, p[a + "w_ks.weight"], p[a + "w_ks.bias"]).view(B, L, H, dk)
    vh = F.linear(bank, p[a + "w_vs.weight"], p[a + "w_vs.bias"]).view(B, L, H, dk)
    s = torch.einsum("bhd,blhd->bhl", qh, kh) / float(np.power(dk, 0.5))
    if mask is not None:
        s = s.masked_fill(mask[:, None, :] == 0.0, float("-inf"))
    pa = torch.softmax(s, dim=2)
    o = torch.einsum("bhl,blhd->bhd", pa, vh).reshape(B, H * dk)
    y = F.linear(o, p[a + "fc.weight"], p[a + "fc.bias"]) + q
    y = layer_norm(y, p[a + "layer_norm.gamma"], p[a + "layer_norm.beta"])
    w1 = p[f + "w_1.weight"].squeeze(-1)
    w2 = p[f + "w_2.weight"].squeeze(-1)
    z = F.linear(F.relu(F.linear(y, w1, p[f + "w_1.bias"])), w2, p[f + "w_2.bias"])
    out = layer_norm(z + y, p[f + "layer_norm.gamma"], p[f + "layer_norm.beta"])
    attn = pa.permute(1, 0, 2).reshape(H * B, 1, L)
    if return_head_diff:
        return out, attn, head_diff(o.view(B, H, dk))
    return out, attn


def head_diff(o):
    """MultiHeadAttention.diff_outputs (submodules.py:38-52), is_regu=True: o [B,H,dv] per-head attention outputs ->
    [B] mean over the ordered head pairs i != j of cos^2(o_i, o_j)  (H = 1: 0 / 0)."""
    x = F.normalize(o, p=2, dim=-1)
    c2 = torch.bmm(x, x.permute(0, 2, 1)) ** 2
    H = o.shape[1]
    idx = torch.arange(H)
    c2[:, idx, idx] = 0
    return c2.sum(dim=[1, 2]) / (H * (H - 1))


def mha_stack(p, stack, q, bank, mask, n_head, d_kv, stack_num):
    """MODEL:509-546: `stack_num` layers, output of layer i is the query of layer i+1."""
    for i in range(stack_num):
        q, _ = sq_mha_layer(p, "%s.%d" % (stack, i), q, bank, mask, n_head, d_kv)
    return q


# ---------------------------------------------------------------------------
# A.1  text-level GCN (max-times aggregation over the n-gram graph)
# ---------------------------------------------------------------------------
def text_gcn_doc(ids, node_hidden, edge_w, pmi, ngram, max_length=100):
    """One document.  Follows Text_GCN.py:142-211 (graph) and 242-275 (max / sum / relu).

    ids: 1-D int sequence (0 = PAD anywhere), node_hidden [V,D] f32 ndarray,
    edge_w [count] f32 ndarray, pmi: object answering pmi[u, v] -> edge id.
    """
    ids = [int(x) for x in ids][:max_length]
    t = [x for x in ids if x != 0]
    D = node_hidden.shape[1]
    best = {}                                    # dst token -> running max [D]
    for i, u in enumerate(t):
        hu = node_hidden[u]
        for j in range(max(0, i - ngram), min(i + ngram + 1, len(t))):
            v = t[j]
            m = np.float32(edge_w[pmi[u, v]]) * hu
            best[v] = m if v not in best else np.maximum(best[v], m)
        m = np.float32(edge_w[pmi[u, u]]) * hu   # the explicit self loop (Text_GCN.py:163-164)
        best[u] = m if u not in best else np.maximum(best[u], m)
    out = np.zeros(D, dtype=np.float32)
    for v in best:                               # PAD node (if any) has no in-edge -> adds 0
        out += best[v]
    return np.maximum(out, 0.0)


def text_gcn(text, node_hidden, edge_w, pmi, ngram, max_length=100):
    """Text_GCN.Model.forward (Text_GCN.py:213-275) for a batch [B,T] -> [B,D] torch f32."""
    text = np.asarray(text)
    nh = node_hidden.detach().numpy() if torch.is_tensor(node_hidden) else np.asarray(node_hidden)
    ew = edge_w.detach().numpy().reshape(-1) if torch.is_tensor(edge_w) else np.asarray(edge_w).reshape(-1)
    out = np.stack([text_gcn_doc(text[b], nh, ew, pmi, ngram, max_length) for b in range(text.shape[0])])
    return torch.from_numpy(out)


# ---------------------------------------------------------------------------
# A.2  text memory bank (embedding + packed 2-layer BiLSTM)
# ---------------------------------------------------------------------------
_LSTM_CACHE = {}


def _lstm_from_params(p, emb_size, hidden, num_layers):
    # keyed on the VALUES (a checksum per LSTM tensor): id(p) alone is reused by the allocator for the next dict
    key = (emb_size, hidden, num_layers) + tuple(float(v.double().sum()) + float(v.double().abs().sum())
                                                 for k, v in sorted(p.items()) if k.startswith("lstm."))
    if key not in _LSTM_CACHE:
        lstm = torch.nn.LSTM(emb_size, hidden, num_layers, bidirectional=True, batch_first=True)
        sd = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
        lstm.load_state_dict(sd)
        lstm.eval()
        _LSTM_CACHE.clear()
        _LSTM_CACHE[key] = lstm
    return _LSTM_CACHE[key]


def text_memory_bank(p, text, text_lens, hidden=150, num_layers=2):
    """MODEL:366-398: embedding (row 0 = 0) -> pack -> BiLSTM -> unpack to T (zeros at pads)."""
    T = text.shape[1]
    emb = F.embedding(text, p["embedding.weight"])
    lstm = _lstm_from_params(p, emb.shape[-1], hidden, num_layers)
    with torch.no_grad():
        packed = torch.nn.utils.rnn.pack_padded_sequence(
            emb, text_lens.cpu(), batch_first=True, enforce_sorted=False)
        out, _ = lstm(packed)
        out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    return out.contiguous()


# ---------------------------------------------------------------------------
# A.6  whole forward (entry at the feature maps)
# ---------------------------------------------------------------------------
def forward(p, inputs, pmi, n_head, d_kv, stack_num, ngram, label_query=None, return_parts=False):
    """Multi_GCN_Multihead_Att.forward (MODEL:431-567) with identity trunks.

    p: params dict (reference state_dict names); inputs: dict of torch CPU tensors with
    keys text, text_lens, text_mask, object_feature [B,2048,14,14], place_feature,
    object_inp [B,80,300], place_inp [B,365,300]; label_query [NLQ,300].
    """
    with torch.no_grad():
        text, lens, mask = inputs["text"], inputs["text_lens"], inputs["text_mask"]
        lq = inputs["label_query"] if label_query is None else label_query
        parts = {}
        text_feature = text_gcn(text.numpy(), p["text_features.node_hidden.weight"],
                                p["text_features.seq_edge_w.weight"], pmi, ngram)
        text_bank = text_memory_bank(p, text, lens)
        att = {}
        bank = {}
        for chan, C_key in (("object", "object_A"), ("place", "place_A")):
            feat = inputs[chan + "_feature"]
            bank[chan] = img_memory_bank(feat, p["liner_img_%s.weight" % chan], p["liner_img_%s.bias" % chan])
            pooled = max_pool(feat)
            G = image_gcn(p[C_key], inputs[chan + "_inp"][0], p["gc1.weight"], p["gc2.weight"])
            x = torch.matmul(pooled, G.transpose(0, 1))
            y = label_attention(p, chan + "_attention", lq, x)
            att[chan] = label_attention_tail(p, chan, y)
            parts[chan + "_x"] = x
            parts[chan + "_att"] = att[chan]
        iot = mha_stack(p, "img_object_text_multi_head_att", att["object"], text_bank, mask, n_head, d_kv, stack_num)
        ipt = mha_stack(p, "img_place_text_multi_head_att", att["place"], text_bank, mask, n_head, d_kv, stack_num)
        tio = mha_stack(p, "text_img_object_multi_head_att", text_feature, bank["object"], None, n_head, d_kv, stack_num)
        tip = mha_stack(p, "text_img_place_multi_head_att", text_feature, bank["place"], None, n_head, d_kv, stack_num)
        multi = torch.cat([tio, tip, iot, ipt], dim=1)
        multi = F.linear(multi, p["multi_linear_1.weight"], p["multi_linear_1.bias"])
        logits = F.linear(multi, p["multi_linear_2.weight"], p["multi_linear_2.bias"])
        if return_parts:
            parts.update(text_feature=text_feature, text_bank=text_bank, tio=tio, tip=tip, iot=iot, ipt=ipt,
                         bank_object=bank["object"], bank_place=bank["place"])
            return logits, parts
        return logits
