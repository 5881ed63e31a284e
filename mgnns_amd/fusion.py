"""Single-query multi-head fusion blocks with the reference's module surface.

Mirrors models/submodules.py (LayerNorm :142-156, MultiHeadAttention :15-94,
PositionwiseFeedForward :122-139) and models/moudles.py (MyMultiHeadAttention :198-230,
MyAnotherMultiHeadAttention :298-324): same class names, constructor arguments, parameter
names and shapes (so reference checkpoints load with strict=True), same return values.
The arithmetic is in libmgnns_hip.so; these classes only hold parameters and sequence kernels.

Eval/forward only: the kernels implement no dropout and no backward, so forward() refuses to
run in training mode instead of silently diverging from the reference.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops


import os as _os

TAIL_TERMS = int(_os.environ.get('MGNNS_TAIL_TERMS', '1'))   # bf16-mode tail: 3 = split-bf16 (hi/lo), 1 = plain bf16
# bf16 mode, 1-term tail: attention core + tail as ONE launch (the tile's last core workgroup runs its tail).  Correct and
# bit-identical (tests), but OFF by default: 100 us per L=196 layer stand-alone against 91 us for the two launches (the in-kernel
# tail has no 4-workgroup cluster for the next projection) and 0.99 against 0.91 ms per forward -- DESIGN.md section 6
FUSED_LAYER = _os.environ.get('MGNNS_FUSED_LAYER', '0') == '1'
# bf16 mode + set_attention('folded'): the composed-map kernels (sq_mha_folded_bf16.hip + mgnns_mha_tail_c16_fwd); 0 = the three
# exact-fp32 launches of sq_mha_folded.hip as in fp32 / bf16x3 mode
FOLDED_BF16 = _os.environ.get('MGNNS_FOLDED_BF16', '1') == '1'
# bf16x3 mode + faithful attention: the split-bf16 core (sq_mha_split_bf16.hip); 0 = the exact-f32 MFMA core of fp32 mode
SPLIT_CORE = _os.environ.get('MGNNS_SPLIT_CORE', '1') == '1'
# ... its masked launches (the text bank) in the GROUPED form: the samples of a group share a workgroup's staging and weight stream
SPLIT_GROUPED = _os.environ.get('MGNNS_SPLIT_GROUPED', '1') == '1'
# split-bf16 (terms = 3) layer tails with fc's K split over a cluster of workgroups like the bf16 mode's
TAIL3_KSPLIT = _os.environ.get('MGNNS_TAIL3_KSPLIT', '1') == '1'


def _require_eval(mod):
    if mod.training:
        raise RuntimeError("%s: mgnns_amd implements the eval-mode forward only (dropout/backward are "
                           "not part of the HIP path); call .eval()" % type(mod).__name__)


class MemoryBank:
    """A key/value memory bank in the forms the kernels consume: fp32 [B,L,D], bf16 [B,L,320]
    (zero padded) and/or the split-bf16 pair [2,B,L,320] (hi + lo images of the fp32 bank).  MyMultiHeadAttention accepts it wherever the reference passes the bank tensor, so a
    bank that feeds several layers is converted once."""

    def __init__(self, f32=None, bf16=None, split=None):
        if f32 is None and bf16 is None and split is None:
            raise ValueError("empty MemoryBank")
        self.f32 = f32
        self._bf16 = bf16
        self._split = split

    @property
    def split(self):
        """bf16 [2, B, L, 320]: the fp32 bank as hi + lo images (the split-bf16 attention core's operand)."""
        if self._split is None:
            if self.f32 is None:
                raise ValueError("the split-bf16 attention needs the fp32 memory bank")
            self._split = ops.split_pad_bf16(self.f32.contiguous())
        return self._split

    @property
    def bf16(self):
        if self._bf16 is None:
            self._bf16 = ops.cast_pad_bf16(self.f32.contiguous())
        return self._bf16

    @property
    def shape(self):
        if self.f32 is not None:
            return self.f32.shape
        return self._bf16.shape if self._bf16 is not None else self._split.shape[1:]


def mask_plan_applies(mask, precision='bf16', attention='faithful'):
    """Does a masked stack in this mode run a packed / grouped attention launch (a plan of the mask makes sense)?  -> False, or the
    kind of plan: 'packed' (bf16 mode: sq_mha32_packed_kernel) | 'grouped' (bf16x3 + faithful: the split-bf16 core's grouped form)."""
    if mask is None or attention != 'faithful' or mask.shape[0] > ops.PLAN_MAX_B:
        return False
    if precision == 'bf16' and ops.MHA_CORE == 32 and ops.MHA_PACKED and mask.shape[-1] <= ops.PLAN_MAX_L and not FUSED_LAYER:
        return 'packed'
    if precision == 'bf16x3' and SPLIT_CORE and SPLIT_GROUPED and mask.shape[-1] <= ops.SPLIT_PLAN_MAX_L:
        return 'grouped'
    return False


def make_mask_plan(mask, precision='bf16', attention='faithful'):
    """The packing plan of a [B, L] attention mask for the masked attention launches of this mode (ops.sq_mha_plan / ops.sq_mha_split_plan:
    the live rows of short samples share a workgroup), or None where it does not apply.  Depends on the mask's VALUES only; whoever
    passes it to run_stack(plan=...) orders the launch that built it in front of the stack (an event if it ran on another stream).
    The model's bf16 forward gets the same plan out of the BiLSTM's prep launch instead (ops.bilstm(plan_mask=...))."""
    kind = mask_plan_applies(mask, precision, attention)
    if not kind:
        return None
    m2 = mask.reshape(mask.shape[0], -1).float().contiguous()
    plan = ops.sq_mha_plan(m2) if kind == 'packed' else ops.sq_mha_split_plan(m2)
    plan._mg_plan_kind = kind
    return plan


class LayerNorm(nn.Module):
    """gamma * (x - mean) / (std_unbiased + eps) + beta  (submodules.py:153-156)."""

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(features))
        self.beta = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        return ops.layernorm(x.contiguous(), self.gamma.detach(), self.beta.detach(), self.eps)


class MultiHeadAttention(nn.Module):
    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1, is_regu=False):
        super().__init__()
        if d_k != d_v:
            raise ValueError("d_k must equal d_v (the reference always passes d_kv for both)")
        self.n_head, self.d_k, self.d_v, self.is_regu = n_head, d_k, d_v, is_regu
        self.precision = 'fp32'      # 'fp32' (exact-f32 MFMA) | 'bf16' (bf16 operands, fp32 accumulate) | 'bf16x3' (split-bf16 tails)
        self.attention = 'faithful'  # 'faithful' (K/V projected as the reference does) | 'folded' (see _folded)
        self._wp = None
        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        nn.init.normal_(self.w_qs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_ks.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_vs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_v)))
        self.layer_norm = LayerNorm(d_model)
        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)
        self.dropout = nn.Dropout(dropout)

    def forward(self, q, k, v, mask=None):
        """q [B,1,d]; k = v = memory bank [B,L,d]; mask [B,1,L] or None -> (out [B,1,d], attn [H*B,1,L])."""
        _require_eval(self)
        if q.dim() != 3 or q.shape[1] != 1:
            raise ValueError("the fusion attention is single-query: q must be [B,1,d], got %s" % (tuple(q.shape),))
        if k is not v and (isinstance(k, MemoryBank) or isinstance(v, MemoryBank) or
                           k.data_ptr() != v.data_ptr() or k.shape != v.shape):
            raise ValueError("key and value must be the same memory bank (as at every reference call site)")
        bank = k if isinstance(k, MemoryBank) else MemoryBank(f32=k.contiguous())
        B = q.shape[0]
        q2 = q.reshape(B, -1).contiguous()
        m2 = None if mask is None else mask.reshape(B, -1).float().contiguous()
        qh = ops.linear(q2, self.w_qs.weight.detach(), self.w_qs.bias.detach())
        if self.attention == 'folded':
            o, attn = self._folded(qh, bank, m2, True)
        elif self.precision == 'bf16':
            o, attn = ops.sq_mha_core_bf16(qh, bank.bf16, m2, self.n_head, self.d_k, self._packed_kv(ops.MHA_CORE_PLAIN),
                                           self.w_ks.bias.detach(), self.w_vs.bias.detach())
        elif self._split_core():
            o, attn = ops.sq_mha_core_split(qh, bank.split, m2, self.n_head, self.d_k, self._packed_kv("split"),
                                            self.w_ks.bias.detach(), self.w_vs.bias.detach())
        else:
            if bank.f32 is None:
                raise ValueError("fp32 attention needs the fp32 memory bank")
            o, attn = ops.sq_mha_core(qh, bank.f32, m2, self.n_head, self.d_k,
                                      self.w_ks.weight.detach(), self.w_ks.bias.detach(),
                                      self.w_vs.weight.detach(), self.w_vs.bias.detach())
        y = ops.linear(o, self.fc.weight.detach(), self.fc.bias.detach(), residual=q2)
        y = self.layer_norm(y)
        if self.is_regu:                    # submodules.py:84-93: the head-difference term as a third result
            return y.view(B, 1, -1), attn, ops.head_diff(o, self.n_head)
        return y.view(B, 1, -1), attn


    def _folded(self, qh, bank, m2, want_attn):
        """K/V projections folded into the query side (csrc/sq_mha_folded.hip): algebraically the reference's
        attention, fp32 throughout, reads whichever copy of the memory bank exists (fp32 preferred)."""
        x = bank.f32 if bank.f32 is not None else bank.bf16
        return ops.sq_mha_folded(qh, x, m2, self.n_head, self.d_k, self.w_ks.weight.detach(),
                                 self.w_vs.weight.detach(), self.w_vs.bias.detach(), want_attn=want_attn)

    def _split_core(self):
        """'bf16x3' + 'faithful': the K/V projections on split-bf16 operands (csrc/sq_mha_split_bf16.hip) -- the reference's
        formulation inside the 1e-4 gate without the exact-f32 MFMA's 16x lower rate (MGNNS_SPLIT_CORE=0: the exact-f32 core)."""
        return (self.precision == 'bf16x3' and self.attention == 'faithful' and SPLIT_CORE and self.d_k == 128
                and self.n_head <= 16 and self.w_ks.in_features <= ops.BANK_LD)

    def _packed_kv(self, form=None):
        """w_ks / w_vs in the MFMA-fragment-major bf16 layout of the attention core's build `form` (ops.MHA_CORE; the fused
        layer kernel takes 16), rebuilt when either weight changes."""
        form = ops.MHA_CORE if form is None else form
        wk, wv = self.w_ks.weight, self.w_vs.weight
        key = (wk.data_ptr(), wk._version, wv.data_ptr(), wv._version, str(wk.device))
        if self._wp is None or self._wp[0] != key:
            ops.retire(self._wp)                # a live capture may hold the old packs' addresses
            self._wp = (key, {})
        packs = self._wp[1]
        if form not in packs:
            if form == "split":
                packs[form] = ops.pack_kv_weights_split(wk.detach(), wv.detach(), self.n_head, self.d_k)
            else:
                packs[form] = ops.pack_kv_weights_bf16(wk.detach(), wv.detach(), self.n_head, self.d_k, form=form)
        return packs[form]


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_in, d_hid, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Conv1d(d_in, d_hid, 1)
        self.w_2 = nn.Conv1d(d_hid, d_in, 1)
        self.layer_norm = LayerNorm(d_in)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        _require_eval(self)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        w1 = self.w_1.weight.detach().view(self.w_1.out_channels, self.w_1.in_channels)
        w2 = self.w_2.weight.detach().view(self.w_2.out_channels, self.w_2.in_channels)
        h = ops.linear(x2, w1, self.w_1.bias.detach(), act=ops.ACT_RELU)
        z = ops.linear(h, w2, self.w_2.bias.detach(), residual=x2)
        return self.layer_norm(z).view(shp)


class MyMultiHeadAttention(nn.Module):
    def __init__(self, n_head, d_model, d_kv, dropout=0.1, need_mask=False, is_regu=False, interaction_type=None):
        super().__init__()
        self.need_mask = need_mask
        self.is_regu = is_regu
        self.interaction_type = interaction_type
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_kv, d_kv, dropout=dropout, is_regu=is_regu)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_model, dropout=dropout)

    def forward(self, q, k, v, mask=None):
        """q [B,d] or [B,1,d]; k,v [B,L,d]; mask [B,L] -> (out [B,d], attn [H*B,1,L])  (moudles.py:207-230)."""
        if q.dim() == 2:
            q = q.unsqueeze(1)
        if mask is not None:
            mask = mask.unsqueeze(1)
        if self.need_mask:
            assert mask is not None, 'Please pass the attention mask to the multi-head'
        if self.is_regu:                    # moudles.py:220-229
            enc_output, enc_slf_attn, head_diff = self.slf_attn(q, k, v, mask)
            enc_output = self.pos_ffn(enc_output)
            return enc_output.squeeze(1), enc_slf_attn, head_diff
        enc_output, enc_slf_attn = self.slf_attn(q, k, v, mask)
        enc_output = self.pos_ffn(enc_output)
        return enc_output.squeeze(1), enc_slf_attn


def _versions(*params):
    return tuple((p.data_ptr(), p._version) for p in params) + (str(params[0].device),)


def _tail_pack(layer):
    """Pre-packed weights of one layer's fused tail, rebuilt when any of them changes."""
    a, f = layer.slf_attn, layer.pos_ffn
    ps = (a.fc.weight, a.fc.bias, a.layer_norm.gamma, a.layer_norm.beta, f.w_1.weight, f.w_1.bias, f.w_2.weight,
          f.w_2.bias, f.layer_norm.gamma, f.layer_norm.beta)
    key = _versions(*ps)
    hit = getattr(layer, "_tail_cache", None)
    if hit is None or hit[0] != key:
        d = {"fc_wp": ops.pack_weight_f32(a.fc.weight.detach()), "fc_b": a.fc.bias.detach(),
             "g1": a.layer_norm.gamma.detach(), "be1": a.layer_norm.beta.detach(),
             "w1_wp": ops.pack_weight_f32(f.w_1.weight.detach().view(f.w_1.out_channels, f.w_1.in_channels)),
             "b1": f.w_1.bias.detach(),
             "w2_wp": ops.pack_weight_f32(f.w_2.weight.detach().view(f.w_2.out_channels, f.w_2.in_channels)),
             "b2": f.w_2.bias.detach(), "g2": f.layer_norm.gamma.detach(), "be2": f.layer_norm.beta.detach()}
        hit = (key, d)
        ops.retire(getattr(layer, "_tail_cache", None))      # a live capture may hold the old pack's addresses
        layer._tail_cache = hit
    return hit[1]


def _tail_pack_bf16(layer):
    a, f = layer.slf_attn, layer.pos_ffn
    ps = (a.fc.weight, a.fc.bias, a.layer_norm.gamma, a.layer_norm.beta, f.w_1.weight, f.w_1.bias, f.w_2.weight,
          f.w_2.bias, f.layer_norm.gamma, f.layer_norm.beta)
    key = _versions(*ps)
    hit = getattr(layer, "_tail_cache_bf16", None)
    if hit is None or hit[0] != key:
        d = {"fc": ops.pack_weight_bf16_split(a.fc.weight.detach()), "fc_b": a.fc.bias.detach(),
             "g1": a.layer_norm.gamma.detach(), "be1": a.layer_norm.beta.detach(),
             "w1": ops.pack_weight_bf16_split(f.w_1.weight.detach().view(f.w_1.out_channels, f.w_1.in_channels)),
             "b1": f.w_1.bias.detach(),
             "w2": ops.pack_weight_bf16_split(f.w_2.weight.detach().view(f.w_2.out_channels, f.w_2.in_channels)),
             "b2": f.w_2.bias.detach(), "g2": f.layer_norm.gamma.detach(), "be2": f.layer_norm.beta.detach()}
        hit = (key, d)
        ops.retire(getattr(layer, "_tail_cache_bf16", None))      # a live capture may hold the old pack's addresses
        layer._tail_cache_bf16 = hit
    return hit[1]


def _wq_pack_bf16(layer):
    a = layer.slf_attn
    key = _versions(a.w_qs.weight, a.w_qs.bias)
    hit = getattr(layer, "_wq_cache_bf16", None)
    if hit is None or hit[0] != key:
        hit = (key, (ops.pack_weight_bf16_split(a.w_qs.weight.detach()), a.w_qs.bias.detach(), a.w_qs.out_features))
        ops.retire(getattr(layer, "_wq_cache_bf16", None))      # a live capture may hold the old pack's addresses
        layer._wq_cache_bf16 = hit
    return hit[1]


def _use_folded_bf16(a):
    """bf16 mode + folded attention: the composed-map form (sq_mha_folded_bf16.hip) instead of the three exact-fp32 launches."""
    return a.attention == 'folded' and a.precision == 'bf16' and not a.is_regu and FOLDED_BF16


def composed_query_map(layer):
    """((hi, lo), bias, H*D) of u_h = (W_k,h^T W_q,h) x + W_k,h^T b_q,h for every head h, stacked [H*D, D]: what a producer of x
    (the previous layer's tail, a channel tail) applies instead of w_qs so that the folded attention starts from u.  Built in
    fp32 (ops.matmul) once per weight version (submodules.py:64-72: q = w_qs(x), k = w_ks(bank))."""
    a = layer.slf_attn
    key = _versions(a.w_qs.weight, a.w_qs.bias, a.w_ks.weight)
    hit = getattr(layer, "_uq_cache", None)
    if hit is None or hit[0] != key:
        H, dk = a.n_head, a.d_k
        wq, bq, wk = a.w_qs.weight.detach(), a.w_qs.bias.detach(), a.w_ks.weight.detach()
        rows, bias = [], []
        for h in range(H):
            wkt = wk[h * dk:(h + 1) * dk].t().contiguous()                                   # [D, dk]
            rows.append(ops.matmul(wkt, wq[h * dk:(h + 1) * dk].contiguous()))               # [D, D]
            bias.append(ops.matmul(wkt, bq[h * dk:(h + 1) * dk].reshape(dk, 1).contiguous()).reshape(-1))
        m = torch.cat(rows, 0).contiguous()
        hit = (key, (ops.pack_weight_bf16_split(m), torch.cat(bias).contiguous(), m.shape[0]), m)
        ops.retire(getattr(layer, "_uq_cache", None))      # a live capture may hold the old pack's addresses
        layer._uq_cache = hit
    return hit[1]


def _composed_query_weight(layer):
    composed_query_map(layer)
    return layer._uq_cache[2]


def _tail_pack_folded(layer):
    """_tail_pack_bf16 with `fc` replaced by fc . blockdiag(W_v) [D, H*D] and its bias by fc b_v + b_fc (submodules.py:74-90:
    the head outputs W_v,h c_h + b_v,h go straight into fc)."""
    a, f = layer.slf_attn, layer.pos_ffn
    ps = (a.fc.weight, a.fc.bias, a.w_vs.weight, a.w_vs.bias, a.layer_norm.gamma, a.layer_norm.beta, f.w_1.weight, f.w_1.bias,
          f.w_2.weight, f.w_2.bias, f.layer_norm.gamma, f.layer_norm.beta)
    key = _versions(*ps)
    hit = getattr(layer, "_tail_cache_folded", None)
    if hit is None or hit[0] != key:
        H, dv = a.n_head, a.d_v
        fc, wv = a.fc.weight.detach(), a.w_vs.weight.detach()
        cols = [ops.matmul(fc[:, h * dv:(h + 1) * dv].contiguous(), wv[h * dv:(h + 1) * dv].contiguous()) for h in range(H)]
        n = torch.cat(cols, 1).contiguous()                                                    # [D, H*D]
        nb = ops.linear(a.w_vs.bias.detach()[None, :].contiguous(), fc.contiguous(), a.fc.bias.detach())[0].contiguous()
        d = dict(_tail_pack_bf16(layer))
        d["fc"] = ops.pack_weight_bf16_split(n)
        d["fc_b"] = nb
        hit = (key, d)
        ops.retire(getattr(layer, "_tail_cache_folded", None))      # a live capture may hold the old pack's addresses
        layer._tail_cache_folded = hit
    return hit[1]


def _tile_counters(layer, B, device):
    """int32 zeros [ceil(B/16)] owned by the layer: arrival counters of the fused layer kernel (it leaves them zero)."""
    n = (B + 15) // 16
    c = getattr(layer, "_tile_counters", None)
    if c is None or c.shape[0] < n or c.device != device:
        c = torch.zeros(max(n, 64), dtype=torch.int32, device=device)
        ops.retire(getattr(layer, "_tile_counters", None))
        layer._tile_counters = c
    return c


def _wq_pack(layer):
    a = layer.slf_attn
    key = _versions(a.w_qs.weight, a.w_qs.bias)
    hit = getattr(layer, "_wq_cache", None)
    if hit is None or hit[0] != key:
        hit = (key, (ops.pack_weight_f32(a.w_qs.weight.detach()), a.w_qs.bias.detach(), a.w_qs.out_features))
        ops.retire(getattr(layer, "_wq_cache", None))
        layer._wq_cache = hit
    return hit[1]


def first_query_pack(layers):
    """(packed fp32 w_qs, bias, H*dk) of a stack's first layer: what a producer kernel needs to emit the stack's first
    projected query itself (ops.label_tail next_q=...), so the stack starts with its attention core."""
    return _wq_pack(list(layers)[0])


def first_query_pack_bf16(layers):
    """The same with the split-bf16 (hi, lo) packed weight, for the bf16-mode producer (ops.label_tail_bf16).  With the folded bf16
    attention the producer applies the composed query map instead (its output feeds sq_mha_folded_bf16 directly)."""
    first = list(layers)[0]
    if _use_folded_bf16(first.slf_attn):
        return composed_query_map(first)
    return _wq_pack_bf16(first)


def first_query(layers, q):
    """What run_stack(..., qh=...) expects for this stack's first layer, computed from q [B, d]: the projected query
    w_qs(q) + b, or the composed query rows when the stack runs the folded bf16 attention."""
    first = list(layers)[0]
    a0 = first.slf_attn
    if _use_folded_bf16(a0):
        _pack, ub, _n = composed_query_map(first)
        return ops.linear(q, _composed_query_weight(first), ub)
    return ops.linear(q, a0.w_qs.weight.detach(), a0.w_qs.bias.detach())


def run_stack(layers, q, bank, mask=None, qh=None, plan=None):
    """A stack of MyMultiHeadAttention layers sharing one memory bank (MODEL:509-546): per layer TWO launches --
    the fused attention core and the fused tail (which also emits the next layer's projected query) -- instead
    of the reference's ~12 small kernels.  Numerically the same chain as calling the layers one by one.
    qh: the first layer's projected query w_qs(q) + b when the producer of q already computed it.
    plan: make_mask_plan(mask) when the caller already built it (both masked stacks of the model share one)."""
    layers = list(layers)
    if not layers:
        return q
    for m in layers:
        _require_eval(m)
        if m.need_mask:
            assert mask is not None, 'Please pass the attention mask to the multi-head'
        if m.is_regu:
            raise NotImplementedError("run_stack drops the head-difference term (the reference's model forward unpacks two values "
                                      "per layer and cannot run with is_regu=True either, MODEL:508-527); call the layers themselves")
    if not isinstance(bank, MemoryBank):
        bank = MemoryBank(f32=bank.contiguous())
    B = q.shape[0]
    q = q.reshape(B, -1).contiguous()
    m2 = None if mask is None else mask.reshape(B, -1).float().contiguous()
    a0 = layers[0].slf_attn
    if _use_folded_bf16(a0):
        # bf16 mode, folded attention: per layer the one-bank-read attention kernel + the tail with the composed maps; qh is the
        # composed query rows u (first_query_pack_bf16 hands the producer the composed map)
        u = qh if qh is not None else first_query(layers, q)
        for i, layer in enumerate(layers):
            a = layer.slf_attn
            c, _ = ops.sq_mha_folded_bf16(u, bank.bf16, m2, a.n_head, a.d_k, want_attn=False)
            nxt = composed_query_map(layers[i + 1]) if i + 1 < len(layers) else None
            q, u = ops.mha_tail_c16(c, q, _tail_pack_folded(layer), a.layer_norm.eps, nxt)
        return q
    if qh is None:
        qh = first_query(layers, q)
    # masked bank: the caller's packing plan of this mask (model.py builds it once per batch), else one for this stack; a plan of
    # another mode's kind (set_precision between building it and running the stack) is dropped
    kind = mask_plan_applies(m2, a0.precision, a0.attention)
    if plan is not None and (not kind or getattr(plan, "_mg_plan_kind", 'packed') != kind):
        plan = None
    if plan is None and kind:
        plan = make_mask_plan(m2, a0.precision, a0.attention)
    for i, layer in enumerate(layers):
        a = layer.slf_attn
        if a.attention == 'folded':
            o, _ = a._folded(qh, bank, m2, False)
        elif a.precision == 'bf16' and FUSED_LAYER and TAIL_TERMS == 1 and a.n_head * a.d_v % 32 == 0:
            # the whole layer in one launch: the tile's last attention-core workgroup runs the tile's tail
            nxt = _wq_pack_bf16(layers[i + 1]) if i + 1 < len(layers) else None
            q, qh = ops.sq_mha_layer_bf16(qh, bank.bf16, m2, a.n_head, a.d_k, a._packed_kv(16), a.w_ks.bias.detach(),
                                          a.w_vs.bias.detach(), q, _tail_pack_bf16(layer), a.layer_norm.eps,
                                          _tile_counters(layer, B, q.device), nxt)
            continue
        elif a.precision == 'bf16':
            # packed masked banks: the 32x32x16 form; one workgroup per sample: whichever form is the faster (ops.MHA_CORE_PLAIN)
            o, _ = ops.sq_mha_core_bf16(qh, bank.bf16, m2, a.n_head, a.d_k,
                                        a._packed_kv(32 if plan is not None else ops.MHA_CORE_PLAIN), a.w_ks.bias.detach(),
                                        a.w_vs.bias.detach(), want_attn=False, plan=plan)
        elif a._split_core():
            o, _ = ops.sq_mha_core_split(qh, bank.split, m2, a.n_head, a.d_k, a._packed_kv("split"), a.w_ks.bias.detach(),
                                         a.w_vs.bias.detach(), want_attn=False, plan=plan)
        else:
            o, _ = ops.sq_mha_core(qh, bank.f32, m2, a.n_head, a.d_k, a.w_ks.weight.detach(), a.w_ks.bias.detach(),
                                   a.w_vs.weight.detach(), a.w_vs.bias.detach(), want_attn=False)
        if a.precision in ('bf16', 'bf16x3') and a.n_head * a.d_v % 32 == 0:
            # bf16 MFMA tail; split-bf16 (hi + lo operands, fp32-class) with MGNNS_TAIL_TERMS=3 and always in 'bf16x3' mode
            nxt = _wq_pack_bf16(layers[i + 1]) if i + 1 < len(layers) else None
            terms = 3 if a.precision == 'bf16x3' else TAIL_TERMS
            # split-bf16 tail: fc's K split over a cluster too (round 5); the next layer's w_qs then is an exact-fp32 GEMM behind it
            nlin = None
            if terms == 3 and nxt is not None and TAIL3_KSPLIT:
                an = layers[i + 1].slf_attn
                nlin = (an.w_qs.weight.detach(), an.w_qs.bias.detach())
            q, qh = ops.mha_tail_bf16(o, q, _tail_pack_bf16(layer), a.layer_norm.eps, nxt, terms=terms,
                                      ksplit=(None if terms == 1 else TAIL3_KSPLIT), next_linear=nlin)
        else:
            nxt = _wq_pack(layers[i + 1]) if i + 1 < len(layers) else None
            q, qh = ops.mha_tail(o, q, _tail_pack(layer), a.layer_norm.eps, nxt)
    return q


class _AnotherMultiHeadAttention(nn.Module):
    """Parameter holder of moudles.py:232-296 (constructed by the reference, never called)."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super().__init__()
        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        self.layer_norm = LayerNorm(d_model)
        self.fc = nn.Linear(n_head * d_v, d_model)


class MyAnotherMultiHeadAttention(nn.Module):
    """Dead branch of the reference (Multi_GCN_Multihead_att.py:517-519,530-532 are commented out);
    kept only so that state_dict keys/shapes match for strict checkpoint loading."""

    def __init__(self, n_head, d_model, d_kv, dropout=0.1, need_mask=False, interaction_type=None):
        super().__init__()
        self.need_mask = need_mask
        self.slf_attn = _AnotherMultiHeadAttention(n_head, d_model, d_kv, d_kv, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_model, dropout=dropout)

    def forward(self, *a, **k):
        raise NotImplementedError("MyAnotherMultiHeadAttention is never called by the reference forward")
