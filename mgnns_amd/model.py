"""Multi_GCN_Multihead_Att with the reference's nn.Module surface
(models/Multi_GCN_Multihead_att.py): same class names, constructor / forward signatures,
parameter names and shapes, so the reference's training script and
engine can construct it, load its checkpoints with strict=True and call
``model(text, text_lens, text_mask, object_img, place_img, object_inp, place_inp)``
(engine/Multi_GCN_Multihead_Att_engine.py:825) unchanged.

All forward arithmetic of the hot path runs in libmgnns_hip.so.  Eval / forward only.
"""
import os
import pickle

import numpy as np
import torch
import torch.nn as nn
from torch.nn import Parameter

from . import _lib, ops
from .adjacency import gen_A, gen_adj_csr
from .fusion import (MemoryBank, MultiHeadAttention, MyAnotherMultiHeadAttention, MyMultiHeadAttention, first_query, make_mask_plan,
                     mask_plan_applies,
                     first_query_pack, first_query_pack_bf16, run_stack)
from .text_gcn import Model as Text_GCN_Model

# Where the packing plan of the text mask (both masked stacks' attention launches) is built, once per batch:
#   'text_gcn' (default, round 5)  one small launch at the END of the text-GCN segment -- a stream with 200 us of slack before the
#              image->text stacks start; they wait for that segment in addition to their own producers;
#   'prep'     an extra workgroup of the BiLSTM's prep launch (ops.bilstm(plan_mask=...)): no launch at all, but the prep launch --
#              the head of the chain the forward follows -- is as long as its slowest workgroup, and the plan workgroup (10-15 us of
#              serial LDS chains) is that: place_bank_first 0.691-0.702 ms two in flight against 0.626-0.628 (NOTES_r05 4);
#   'tails'    round 4: one launch per channel at the head of its label-attention tail segment.
PLAN_SITE = os.environ.get("MGNNS_PLAN_SITE", "prep" if os.environ.get("MGNNS_PLAN_IN_PREP") == "1" else "text_gcn")
if PLAN_SITE not in ("text_gcn", "prep", "tails"):
    raise ValueError("MGNNS_PLAN_SITE must be text_gcn | prep | tails")
PLAN_IN_PREP = PLAN_SITE == "prep"

LABEL_GLOVE_CANDIDATES = ('data/glove/tumblr_label_glove.pkl', 'data/tumblr_label_glove.pkl')


class GraphConvolution(nn.Module):
    """support = X W (dense, MFMA) then adj @ support (MODEL:30-63).  `adj` may be the dense
    normalised adjacency (it is turned into CSR on the fly) or a (row_ptr, col, val) CSR triple."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.Tensor(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.Tensor(1, 1, out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / np.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj, act=ops.ACT_NONE):
        support = ops.matmul(input.float().contiguous(), self.weight.detach())
        if torch.is_tensor(adj):
            adj = ops.dense_to_csr(adj.contiguous())
        # bias=True (never built by the reference model, MODEL:40-41,55-56): output + bias in the propagation's epilogue
        return ops.spmm_csr(adj, support, act=act, bias=None if self.bias is None else self.bias.detach())

    def __repr__(self):
        return '%s (%d -> %d)' % (self.__class__.__name__, self.in_features, self.out_features)


class Attention(nn.Module):
    """Label-GloVe query x image-GCN scores "attention" (MODEL:65-133)."""

    def __init__(self, hid_dim, image_dim, n_heads, dropout):
        super().__init__()
        self.hid_dim = hid_dim
        self.n_heads = n_heads
        assert hid_dim % n_heads == 0
        self.w_q = nn.Linear(hid_dim, hid_dim)
        self.w_k = nn.Linear(image_dim, hid_dim)
        self.w_v = nn.Linear(image_dim, hid_dim)
        self.fc = nn.Linear(hid_dim, hid_dim)
        self.do = nn.Dropout(dropout)

    def forward(self, query, key, value, mask=None):
        """query [NLQ,hid] (label GloVe, any float dtype), key = value [B,image_dim] -> [B,NLQ,hid]."""
        if self.training:
            raise RuntimeError("Attention: eval-mode forward only on the HIP path; call .eval()")
        if key.data_ptr() != value.data_ptr():
            raise ValueError("key and value must be the same tensor (as at MODEL:476,503)")
        Q = ops.linear(query.float().contiguous(), self.w_q.weight.detach(), self.w_q.bias.detach())
        K = ops.linear(key.contiguous(), self.w_k.weight.detach(), self.w_k.bias.detach())
        V = ops.linear(key.contiguous(), self.w_v.weight.detach(), self.w_v.bias.detach())
        x = ops.label_attn_core(Q, K, V, self.n_heads, mask=mask)      # mask: MODEL:118-119 (no reference call site passes one)
        return ops.linear(x, self.fc.weight.detach(), self.fc.bias.detach())


class _NoTrunk(nn.Module):
    """Placeholder when no CNN trunk is supplied: the model then only accepts precomputed
    [B,2048,h,w] feature maps (the benchmark entry)."""

    def forward(self, x):
        raise RuntimeError("no CNN trunk was given to Multi_GCN_Multihead_Att; pass [B,2048,14,14] feature maps")


def _trunk(m):
    """MODEL:274-294: Sequential(conv1, bn1, relu, maxpool, layer1..4) of the given ResNet -- same child indices, so the
    same state_dict keys -- with the HIP forward of mgnns_amd.trunk (row f4)."""
    if m is None:
        return _NoTrunk()
    from .trunk import ResNetFeatures
    return ResNetFeatures(m)


class _PlanCtx(dict):
    """Results of a forward plan's segments by name; prepare(): what must exist before the first segment runs (nothing, or the
    split classifier head's buffers)."""

    def prepare(self):
        pass


def default_schedule(batch, precision='fp32'):
    """What schedule 'auto' resolves to (Multi_GCN_Multihead_Att.resolve_schedule): chosen from alternating-run A/Bs (NOTES_r04 / r05),
    pinned by tests/test_surface_cpu.py so that a default cannot change silently.  'bf16x3': 'channels2' at every batch; otherwise
    'place_bank_first' from 128 samples, 'channels2' from 64, 'small' below."""
    if precision == 'bf16x3':
        return 'channels2'
    return 'place_bank_first' if batch >= 128 else ('channels2' if batch >= 64 else 'small')


class Multi_GCN_Multihead_Att(nn.Module):
    def __init__(self, opt, num_labels, text_model, object_model, place_model,
                 object_num_classes, place_num_classes, object_t=0, place_t=0, in_channel=300,
                 object_adj_file=None, place_adj_file=None, label_glove=None):
        super().__init__()
        self.emb_path = opt['emb_path']
        self.bidirectional = opt['bidirectional']
        self.num_directions = 2 if self.bidirectional else 1
        self.hidden_size = opt['hidden_size']
        self.bi_hidden_size = self.num_directions * opt['hidden_size']
        opt['bi_hidden_size'] = self.bi_hidden_size
        self.d_model = self.bi_hidden_size
        self.pad_idx = 0
        self.stack_num = opt['stack_num']
        self.n_head = opt['n_head']
        self.d_kv = opt['d_kv']
        self.is_regu = opt['is_regu']

        self.embedding = nn.Embedding(opt['vocab_size'], opt['emb_size'], padding_idx=self.pad_idx)
        self.init_weights(opt['emb_type'], self.pad_idx)
        rnn_kw = dict(input_size=opt['emb_size'], hidden_size=opt['hidden_size'], num_layers=opt['num_layers'],
                      bidirectional=opt['bidirectional'], batch_first=True, dropout=opt['dropout'])
        self.rnn = nn.GRU(**rnn_kw)          # constructed, never used (MODEL:172-177,377)
        self.lstm = nn.LSTM(**rnn_kw)
        self.object_gate = nn.Linear(self.bi_hidden_size * 2, self.bi_hidden_size)   # dead (MODEL:548-556)
        self.place_gate = nn.Linear(self.bi_hidden_size * 2, self.bi_hidden_size)

        def stack(need_mask, kind):
            return nn.ModuleList([MyMultiHeadAttention(self.n_head, self.d_model, self.d_kv, dropout=opt['dropout'],
                                                       need_mask=need_mask, is_regu=self.is_regu, interaction_type=kind)
                                  for _ in range(self.stack_num)])

        self.img_object_text_multi_head_att = stack(True, 'img_object_text')
        self.text_object_text_multi_head_att = MyAnotherMultiHeadAttention(
            self.n_head, self.d_model, self.d_kv, dropout=opt['dropout'], need_mask=False,
            interaction_type='text_object_text')
        self.img_place_text_multi_head_att = stack(True, 'img_place_text')
        self.text_place_text_multi_head_att = MyAnotherMultiHeadAttention(
            self.n_head, self.d_model, self.d_kv, dropout=opt['dropout'], need_mask=False,
            interaction_type='text_place_text')
        self.text_img_object_multi_head_att = stack(False, 'text_img_object')
        self.text_img_place_multi_head_att = stack(False, 'text_img_place')

        self.liner_img_object = nn.Linear(2048, self.bi_hidden_size)
        self.liner_img_place = nn.Linear(2048, self.bi_hidden_size)

        self.text_features = text_model
        self.object_features = _trunk(object_model)
        self.place_features = _trunk(place_model)

        self.num_labels = num_labels
        self.object_num_classes = object_num_classes
        self.place_num_classes = place_num_classes
        self.object_t = object_t
        self.place_t = place_t

        self.pooling = nn.MaxPool2d(14, 14)   # kept for the surface; the pool is fused into the bank kernel
        self.gc1 = GraphConvolution(in_channel, 1024)
        self.gc2 = GraphConvolution(1024, 2048)
        self.leakyrelu = nn.LeakyReLU(0.2)
        self.tanh = nn.Tanh()
        self.relu = nn.ReLU()
        self.object_attention = Attention(hid_dim=300, image_dim=self.object_num_classes, n_heads=5, dropout=0.5)
        self.place_attention = Attention(hid_dim=300, image_dim=self.place_num_classes, n_heads=5, dropout=0.5)

        self.object_linear_1 = nn.Linear(2048, 1024)    # dead
        self.object_linear_2 = nn.Linear(1024, 512)     # dead
        self.object_linear_3 = nn.Linear(512, 256)      # dead
        self.object_linear_5 = nn.Linear(300, 100)
        self.object_x_linear = nn.Linear(700, 300)
        self.place_linear_1 = nn.Linear(2048, 1024)     # dead
        self.place_linear_2 = nn.Linear(1024, 512)      # dead
        self.place_linear_3 = nn.Linear(512, 256)       # dead
        self.place_linear_5 = nn.Linear(300, 100)
        self.place_x_linear = nn.Linear(700, 300)
        self.dropout = nn.Dropout(0.5)
        self.multi_linear_1 = nn.Linear(1200, self.bi_hidden_size)
        self.multi_linear_2 = nn.Linear(self.bi_hidden_size, num_labels)

        self.object_A = Parameter(self._init_A(object_num_classes, object_t, object_adj_file))
        self.place_A = Parameter(self._init_A(place_num_classes, place_t, place_adj_file))

        self.image_normalization_mean = [0.485, 0.456, 0.406]
        self.image_normalization_std = [0.229, 0.224, 0.225]

        # label-GloVe query: a module-level pickle load in the reference (MODEL:19-27)
        self.register_buffer('label_query', None, persistent=False)
        self._load_label_query(label_glove if label_glove is not None else opt.get('label_glove'))
        self._wt_cache = {}
        self._lstm_cache = ops.LstmCache()     # derived LSTM weight forms live and die with this module
        self._streams = None
        self.use_streams = bool(opt.get('use_streams', True))
        # 'auto': 'place_bank_first' for batches of at least 128 samples, 'channels2' below (resolve_schedule)
        self.schedule = opt.get('schedule', os.environ.get('MGNNS_SCHEDULE', 'auto'))
        # the classifier as four shares behind the four stacks instead of a segment of its own (forward_plan)
        self.split_head = os.environ.get('MGNNS_SPLIT_HEAD', '1') == '1'
        self.fused_label_tail = os.environ.get('MGNNS_FUSED_LABEL_TAIL', '1') == '1'
        self.fused_label_tail_min_batch = int(os.environ.get('MGNNS_FUSED_TAIL_MIN_BATCH', '96'))      # fp32 fused tail (16 CUs per 256 samples)
        # the bf16 fused tail runs as 4-workgroup clusters: it wins at every batch size (B=32: 0.471 -> 0.431 ms, B=16: 0.438 -> 0.419)
        self.fused_label_tail_bf16_min_batch = int(os.environ.get('MGNNS_FUSED_TAIL_BF16_MIN_BATCH', '1'))
        self.fused_label_tail_bf16 = os.environ.get('MGNNS_FUSED_LABEL_TAIL_BF16', '1') == '1'
        self.fused_head = os.environ.get('MGNNS_FUSED_HEAD', '1') == '1'      # classifier as one launch (composed maps)
        self.fused_label_gcn = os.environ.get('MGNNS_FUSED_LABEL_GCN', '1') == '1'      # label GCN as one persistent launch
        self.label_tail_terms = int(os.environ.get('MGNNS_LABEL_TAIL_TERMS', '3'))
        self.precision = 'fp32'
        self.attention_choice = 'faithful'
        self.attention = 'faithful'
        self.set_precision(opt.get('precision', 'fp32'))
        self.set_attention(opt.get('attention', 'faithful'))

    def set_precision(self, precision):
        """'fp32': every contraction on the exact-f32 MFMA (the parity path, <=1e-4 on logits).
        'bf16' (BASELINE configs[2]): bf16 operands with fp32 accumulation in the image-bank projection, the fusion
        attention's K/V projections, the fused layer tail (plain bf16 weights; MGNNS_TAIL_TERMS=3 selects split hi+lo
        weights) and the BiLSTM's input and recurrent products (MGNNS_LSTM_REC=f32 keeps the exact fp32 recurrence);
        text GCN, label GCN / attention, scores, softmax, LayerNorms, gates, cell state and residuals stay fp32.
        Measured |logit - fp32 CPU oracle| at B=256: 1.6e-2 (not inside the 1e-4 gate, which is the fp32 mode's).
        A feature map whose position count P is not in (104, 200] or not a multiple of 4 uses the fp32 bank kernel
        for that projection in either mode (the bf16 bank kernel is tiled for 14x14 maps); results then carry fp32
        accuracy, never less.
        'bf16x3' (split-bf16): the parity-grade mode that is not 8x slower.  Every fp32 operand of the heavy products is carried
        as bf16 hi + lo and a product is three bf16 MFMAs (~2^-16 relative, fp32 accumulation): the image-bank projection
        (csrc/imgbank_split.hip), the label GCN, the channel tails and the fusion layers' tails; the text bank runs the exact
        fp32 LSTM; scores, softmax, LayerNorms, residuals fp32.  The fusion attention's K/V projection has no split form (the
        bank's hi + lo images for 196 positions are 263 KB, LDS holds 160): with set_attention('folded') -- the pairing
        bench.py reports -- the attention is the exact-fp32 folded kernel; with 'faithful' it is the exact-f32 MFMA core."""
        if precision not in ('fp32', 'bf16', 'bf16x3'):
            raise ValueError("precision must be 'fp32', 'bf16' or 'bf16x3'")
        self.precision = precision
        for m in self.modules():
            if isinstance(m, MultiHeadAttention):
                m.precision = precision
        return self.set_attention(self.attention_choice)

    def set_attention(self, attention):
        """'faithful' (default): the fusion attention projects K and V from the memory bank as the reference does
        (submodules.py:64-72; the MFMA kernels the north-star's utilisation figure is quoted on, and what bench.py's headline runs).
        'folded': the same attention with the projections folded away algebraically (len_q == 1: q.(W_k x_l) = (W_k^T q).x_l and
        sum_l p_l W_v x_l = W_v sum_l p_l x_l) -- one read of the memory bank instead of 242 MFLOP per sample and layer, same results
        to rounding (tests: the reference's goldens); bench.py reports it as a variant.  In bf16 mode the query / output projections
        are composed with them too (csrc/sq_mha_folded_bf16.hip + the c16 tail): the fastest forward of this library (457 k against
        371 k samples/s at B=256).  In fp32 / bf16x3 mode it is the exact-fp32 kernel of csrc/sq_mha_folded.hip.
        'auto': 'folded' in the bf16 modes, 'faithful' in fp32 mode."""
        if attention not in ('auto', 'faithful', 'folded'):
            raise ValueError("attention must be 'auto', 'faithful' or 'folded'")
        self.attention_choice = attention
        self.attention = attention if attention != 'auto' else ('faithful' if self.precision == 'fp32' else 'folded')
        for m in self.modules():
            if isinstance(m, MultiHeadAttention):
                m.attention = self.attention
        return self

    # ---- construction helpers -------------------------------------------------------------------
    @staticmethod
    def _init_A(num_classes, t, adj_file):
        if adj_file is None:          # weights arrive through load_state_dict
            return torch.eye(num_classes)
        adj, _ = gen_A(num_classes, t, adj_file)
        return torch.from_numpy(adj).float()

    def _load_label_query(self, src):
        if torch.is_tensor(src) or isinstance(src, np.ndarray):
            self.label_query = torch.as_tensor(src).float()
            return
        paths = [src] if src else list(LABEL_GLOVE_CANDIDATES)
        for p in paths:
            if p and os.path.exists(p):
                with open(p, 'rb') as f:
                    self.label_query = torch.from_numpy(np.array(pickle.load(f))).float()
                return

    def set_label_query(self, q):
        self.label_query = torch.as_tensor(q).float().to(self.embedding.weight.device)

    def init_weights(self, emb_type, pad_idx):
        if emb_type == 'random':
            self.embedding.weight.data.uniform_(-0.1, 0.1)
        else:
            with open(self.emb_path, 'rb') as f:
                weights = pickle.load(f)
            self.embedding.weight.data = torch.Tensor(weights)
        self.embedding.weight.data[pad_idx] = 0

    # ---- forward pieces ----------------------------------------------------------------------------
    def _lstm_weights(self):
        ws = []
        for layer in range(self.lstm.num_layers):
            for suffix in ("", "_reverse"):
                ws.append(tuple(getattr(self.lstm, "%s_l%d%s" % (n, layer, suffix)).detach()
                                for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
        return ws

    def _text_bank(self, text, text_lens, plan_mask=None):
        """MemoryBank of the text (fp32 + bf16 copy in bf16 mode, both written by the LSTM kernel).  plan_mask: the text mask
        whose packing plan the bf16 prep launch builds on the side (bank.mask_plan; None where it cannot)."""
        lens = text_lens.to(device=text.device, dtype=torch.int64, non_blocking=True).contiguous()
        if self.precision == 'bf16':                        # ('bf16x3': the exact fp32 recurrence below)
            rec = os.environ.get("MGNNS_LSTM_REC", "bf16")
            if plan_mask is not None and not ops.bilstm_can_plan(text.shape[0], text.shape[1], self.embedding.weight.shape[1], rec):
                plan_mask = None
            r = ops.bilstm(text.long().contiguous(), lens, self.embedding.weight.detach(), self._lstm_weights(),
                           self.hidden_size, self.lstm.num_layers, want_bf16=True, recurrence=rec, cache=self._lstm_cache,
                           plan_mask=plan_mask)
            bank = MemoryBank(f32=r[0], bf16=r[1])
            bank.mask_plan = r[2] if plan_mask is not None else None     # the text mask's packing plan, built by the prep launch
            return bank
        return MemoryBank(f32=ops.bilstm(text.long().contiguous(), lens, self.embedding.weight.detach(),
                                         self._lstm_weights(), self.hidden_size, self.lstm.num_layers,
                                         cache=self._lstm_cache))

    def get_text_memory_bank(self, text, text_lens, return_last_state=True):
        """Embedding gather + packed 2-layer BiLSTM + re-padding to T (MODEL:366-398) as HIP kernels.
        Returns (memory_bank [B,T,2*hidden], final state of the last layer [B,2*hidden]) like the reference;
        the final state is read back out of the bank (forward: position len-1, reverse: position 0)."""
        if not self.bidirectional:
            raise NotImplementedError("the HIP text bank implements the bidirectional LSTM the reference configures")
        batch_size, max_text_len = list(text.size())
        lens = text_lens.to(device=text.device, dtype=torch.int64, non_blocking=True).contiguous()
        memory_bank = ops.bilstm(text.long().contiguous(), lens, self.embedding.weight.detach(),
                                 self._lstm_weights(), self.hidden_size, self.lstm.num_layers, cache=self._lstm_cache)
        assert memory_bank.size() == torch.Size([batch_size, max_text_len, self.bi_hidden_size])
        if not return_last_state:
            return memory_bank
        H = self.hidden_size
        idx = (lens - 1).clamp(min=0)
        fwd_last = memory_bank[torch.arange(batch_size, device=text.device), idx, :H]
        bwd_last = memory_bank[:, 0, H:]
        # reference order: cat(enc_final_state[-1] (reverse), enc_final_state[-2] (forward)) (MODEL:392)
        return memory_bank, torch.cat((bwd_last, fwd_last), 1)

    def _wt(self, lin):
        """Linear(2048, 300).weight transposed + padded for the bank kernel, cached per weight version."""
        w = lin.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = self._wt_cache.get(id(lin))
        if hit is None or hit[0] != key:
            hit = (key, ops.transpose_pad(w.detach().contiguous(), ops.IMGBANK_LDW))
            self._cache_put(id(lin), hit)
        return hit[1]

    def _wp(self, lin):
        """The same weight in the MFMA-fragment-major bf16 layout of the bf16 bank kernel."""
        w = lin.weight
        key = (w.data_ptr(), w._version, str(w.device), 'bf16')
        hit = self._wt_cache.get((id(lin), 'bf16'))
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_imgbank_weights_bf16(w.detach().contiguous()))
            self._cache_put((id(lin), 'bf16'), hit)
        return hit[1]

    def _wp_split(self, lin):
        """The same weight as split-bf16 (hi, lo) fragment-major buffers for the bf16x3 bank kernel."""
        w = lin.weight
        key = (w.data_ptr(), w._version, str(w.device), 'split')
        hit = self._wt_cache.get((id(lin), 'split'))
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_weight_bf16_split(w.detach().contiguous()))
            self._cache_put((id(lin), 'split'), hit)
        return hit[1]

    def _img_bank_and_pool(self, feats, lin):
        """-> (MemoryBank, pooled [B,2048]); one pass over the feature map."""
        B = feats.shape[0]
        f3 = feats.float().contiguous().view(B, feats.shape[1], -1)
        if self.precision == 'bf16' and 16 <= f3.shape[2] <= 208 and f3.shape[2] % 4 == 0 and f3.shape[1] % 64 == 0 \
                and lin.out_features <= 304:        # (the bf16 bank kernels' limits; other shapes take the fp32 kernel)
            keep_halves = (self.fused_label_tail and self.fused_label_tail_bf16 and B >= self.fused_label_tail_bf16_min_batch)
            bank, pooled = ops.imgbank_pool_bf16(f3, self._wp(lin), lin.bias.detach(), lin.out_features, combine=not keep_halves)
            return MemoryBank(bf16=bank), pooled
        if self.precision == 'bf16x3' and f3.shape[2] % 4 == 0 and f3.shape[2] <= 224 and f3.shape[1] % 64 == 0 \
                and lin.out_features <= 304:
            consumers = (self.text_img_object_multi_head_att if lin is self.liner_img_object else self.text_img_place_multi_head_att)
            if lin.out_features % 2 == 0 and len(consumers) and all(m.slf_attn._split_core() for m in consumers):
                # the split-bf16 attention core's operand straight from the bank kernel: no fp32 bank, no conversion pass
                _, pooled, sp = ops.imgbank_pool_split(f3, self._wp_split(lin), lin.bias.detach(), lin.out_features,
                                                       want_f32=False, want_split=True)
                return MemoryBank(split=sp), pooled
            bank, pooled = ops.imgbank_pool_split(f3, self._wp_split(lin), lin.bias.detach(), lin.out_features)
            return MemoryBank(f32=bank), pooled            # pooled: [B, 2, K] region-half maxima (the fused tail combines them)
        bank, pooled = ops.imgbank_pool(f3, self._wt(lin), lin.bias.detach(), lin.out_features)
        return MemoryBank(f32=bank), pooled

    def get_img_object_memory_bank(self, img_object_feats):
        """[B,2048,14,14] -> [B,196,300] fp32 (MODEL:400-416)."""
        f3 = img_object_feats.float().contiguous().view(img_object_feats.shape[0], img_object_feats.shape[1], -1)
        return ops.imgbank_pool(f3, self._wt(self.liner_img_object), self.liner_img_object.bias.detach(),
                                self.liner_img_object.out_features, want_pool=False)[0]

    def get_img_place_memory_bank(self, img_place_feats):
        f3 = img_place_feats.float().contiguous().view(img_place_feats.shape[0], img_place_feats.shape[1], -1)
        return ops.imgbank_pool(f3, self._wt(self.liner_img_place), self.liner_img_place.bias.detach(),
                                self.liner_img_place.out_features, want_pool=False)[0]

    def _label_gcn(self, A, inp):
        """MODEL:460-473 / 489-499: gen_adj + GraphConvolution x2 over the label graph -> G [C, 2048] (batch independent)."""
        _, csr = gen_adj_csr(A)
        x = self.gc1(inp[0].float().contiguous(), csr, act=ops.ACT_LRELU2)
        G = self.gc2(x, csr)
        ops.stamp("  label GCN end")
        return G

    def _cache_put(self, key, value):
        """Replace a derived-weights cache entry.  While a captured hipGraph of this model exists (GraphedForward counts
        itself in _live_graphs) the superseded entry is parked, not freed: the graph holds raw addresses of its buffers --
        packed weights, the persistent launches' scratch with their queue counters -- and a replay after set_precision() /
        load_state_dict() must not read recycled memory.  The park empties when the last graph goes."""
        old = self._wt_cache.get(key)
        if old is not None and getattr(self, "_live_graphs", 0) > 0:
            if not hasattr(self, "_wt_retired"):
                self._wt_retired = []
            self._wt_retired.append(old)
        self._wt_cache[key] = value

    def _tail_pack(self, attention, linear_5, x_linear):
        """Packed weights of one channel's fused label-attention tail, rebuilt when any of them changes:
        w_k / w_v / x_linear in the fragment-major fp32 layout, fc and linear_5 composed into one map (no non-linearity
        between them, MODEL:131 -> 477): Wc = W5 . Wfc, bc = W5 . b_fc + b5."""
        ps = (attention.w_k.weight, attention.w_k.bias, attention.w_v.weight, attention.w_v.bias, attention.fc.weight,
              attention.fc.bias, linear_5.weight, linear_5.bias, x_linear.weight, x_linear.bias)
        key = tuple((p_.data_ptr(), p_._version) for p_ in ps) + (str(ps[0].device),)
        hit = self._wt_cache.get((id(attention), 'tail'))
        if hit is None or hit[0] != key:
            w5 = linear_5.weight.detach()
            wc = ops.matmul(w5.contiguous(), attention.fc.weight.detach().contiguous())               # [N5, hid]
            bc = ops.linear(attention.fc.bias.detach()[None, :].contiguous(), w5, linear_5.bias.detach())[0]
            d = {"wk": ops.pack_weight_f32(attention.w_k.weight.detach()), "bk": attention.w_k.bias.detach(),
                 "wv": ops.pack_weight_f32(attention.w_v.weight.detach()), "bv": attention.w_v.bias.detach(),
                 "wc": ops.pack_weight_f32(wc), "bc": bc.contiguous(), "n5": linear_5.out_features,
                 "xl": ops.pack_weight_f32(x_linear.weight.detach()), "bxl": x_linear.bias.detach(),
                 "n_out": x_linear.out_features, "C": attention.w_k.in_features, "_src": ps}
            hit = (key, d)
            self._cache_put((id(attention), 'tail'), hit)
        return hit[1]

    def _lgcn_pack(self, tag):
        """Packed GraphConvolution weights of one channel for the persistent label-GCN launch (exact fp32 fragments in fp32
        mode, split-bf16 pairs in bf16 mode), rebuilt when a weight or the precision changes; carries the launch's scratch."""
        gc1, gc2 = (self.gc1, self.gc2)
        ps = (gc1.weight, gc2.weight)
        split = self.precision in ('bf16', 'bf16x3')
        key = tuple((p_.data_ptr(), p_._version) for p_ in ps) + (str(ps[0].device), split)
        hit = self._wt_cache.get(('lgcn', tag))
        if hit is None or hit[0] != key:
            d = ops.label_gcn_pack(gc1.weight.detach(), gc2.weight.detach(), split)
            d["_src"] = ps
            hit = (key, d)
            self._cache_put(('lgcn', tag), hit)
        return hit[1]

    def _head_pack(self):
        """multi_linear_2 . multi_linear_1 as one [num_labels, 1200] map (MODEL:563-566: only dropout between them, the
        identity in eval), rebuilt when either changes: Wc = W2 . W1, bc = W2 . b1 + b2."""
        l1, l2 = self.multi_linear_1, self.multi_linear_2
        ps = (l1.weight, l1.bias, l2.weight, l2.bias)
        key = tuple((p_.data_ptr(), p_._version) for p_ in ps) + (str(ps[0].device),)
        hit = self._wt_cache.get('head')
        if hit is None or hit[0] != key:
            wc = ops.matmul(l2.weight.detach().contiguous(), l1.weight.detach().contiguous())                # [NL, 1200]
            bc = ops.linear(l1.bias.detach()[None, :].contiguous(), l2.weight.detach(), l2.bias.detach())[0]
            hit = (key, (wc.contiguous(), bc.contiguous(), ps))
            self._cache_put('head', hit)
        return hit[1][0], hit[1][1]

    def _tail_pack_bf16(self, attention, linear_5, x_linear):
        """The same weights as split-bf16 fragment-major (hi, lo) buffers for the bf16-mode fused tail."""
        ps = (attention.w_k.weight, attention.w_k.bias, attention.w_v.weight, attention.w_v.bias, attention.fc.weight,
              attention.fc.bias, linear_5.weight, linear_5.bias, x_linear.weight, x_linear.bias)
        key = tuple((p_.data_ptr(), p_._version) for p_ in ps) + (str(ps[0].device),)
        hit = self._wt_cache.get((id(attention), 'tail_bf16'))
        if hit is None or hit[0] != key:
            w5 = linear_5.weight.detach()
            wc = ops.matmul(w5.contiguous(), attention.fc.weight.detach().contiguous())
            bc = ops.linear(attention.fc.bias.detach()[None, :].contiguous(), w5, linear_5.bias.detach())[0]
            sp = lambda w: ops.pack_weight_bf16_split(w.contiguous())
            d = {"wk": sp(attention.w_k.weight.detach()), "bk": attention.w_k.bias.detach(),
                 "wv": sp(attention.w_v.weight.detach()), "bv": attention.w_v.bias.detach(),
                 "wc": sp(wc), "bc": bc.contiguous(), "n5": linear_5.out_features,
                 "xl": sp(x_linear.weight.detach()), "bxl": x_linear.bias.detach(),
                 "n_out": x_linear.out_features, "C": attention.w_k.in_features, "_src": ps}
            hit = (key, d)
            self._cache_put((id(attention), 'tail_bf16'), hit)
        return hit[1]

    def _lgcn_fused_ok(self, C, K0):
        """Does the persistent label-GCN launch take this channel?  (mgnns_label_gcn_supported: the launcher's own limits)"""
        return bool(_lib.lib().mgnns_label_gcn_supported(int(C), int(K0), self.gc1.out_features, self.gc2.out_features,
                                                         1 if self.precision in ('bf16', 'bf16x3') else 0))

    def _label_q(self, attention):
        """w_q(label query) [NLQ, hid] (MODEL:97): batch independent, computed next to the label GCN."""
        return ops.linear(self.label_query.float().contiguous(), attention.w_q.weight.detach(), attention.w_q.bias.detach())

    def _channel_tail(self, pooled, G, Gp, Q, attention, linear_5, x_linear, next_stack=None):
        """Second half of an image channel (MODEL:474-479 / 500-506): read-out through the label GCN, then label attention
        + 300->100->700->300 + the projected query of the fusion stack the feature feeds as ONE fused launch
        (csrc/label_tail.hip).  bf16 mode: always (four-workgroup clusters per 16 samples).  fp32 mode: for batches of at
        least fused_label_tail_min_batch samples -- that kernel runs on B/16 CUs, which wins when the small launches it
        replaces would each queue for a CU behind the chip-filling kernels (B=256: 0.955 vs 0.993 ms per forward) and loses
        on an idle chip (B=32: 0.61 vs 0.54 ms).  Below the threshold, or with MGNNS_FUSED_LABEL_TAIL=0: the chain of
        module-level operators.  -> (att [B,300], qh or None)"""
        B = pooled.shape[0]
        L = _lib.lib()
        n_next = 1 if next_stack is not None and len(next_stack) else 0
        C, NLQ, nh = attention.w_k.in_features, Q.shape[0], attention.n_heads
        dh, N5, n_out = Q.shape[1] // nh, linear_5.out_features, x_linear.out_features
        flat_ok = x_linear.in_features == NLQ * N5
        # the fused launches have shape limits (their argument checks, exported as predicates): other shapes -- e.g. more than
        # 384 label classes in bf16 mode -- run the chain of module-level operators below
        bf16_ok = (Gp is not None and flat_ok and pooled.dim() == 3 and
                   L.mgnns_label_tail_bf16_supported(C, NLQ, nh, dh, N5, n_out, pooled.shape[-1], int(self.label_tail_terms), n_next))
        f32_ok = flat_ok and L.mgnns_label_tail_supported(C, NLQ, nh, dh, N5, n_out, 0, n_next)
        if bf16_ok and self.fused_label_tail and B >= self.fused_label_tail_bf16_min_batch:
            # bf16 precision mode: the whole chain, read-out included, as ONE launch on the bf16 MFMA
            pk = self._tail_pack_bf16(attention, linear_5, x_linear)
            nq = first_query_pack_bf16(next_stack) if next_stack is not None and len(next_stack) else None
            parts = pooled if pooled.dim() == 3 else pooled.unsqueeze(1)
            r = ops.label_tail_bf16(parts.contiguous(), Gp, Q, attention.n_heads, pk, next_q=nq, terms=self.label_tail_terms)
            return r if nq is not None else (r, None)
        if pooled.dim() == 3:
            pooled = pooled.amax(dim=1) if pooled.shape[1] > 1 else pooled[:, 0].contiguous()
        if self.fused_label_tail and B >= self.fused_label_tail_min_batch and f32_ok:
            pk = self._tail_pack(attention, linear_5, x_linear)
            nq = first_query_pack(next_stack) if next_stack is not None and len(next_stack) else None
            # the read-out stays a launch of its own: 2*B*2048*C FLOPs on the exact-f32 MFMA want many CUs, the fused
            # kernel runs on B/16 of them (measured with the read-out inside: 0.99 vs 0.955 ms per B=256 forward)
            r = ops.label_tail(ops.linear(pooled.contiguous(), G), Q, attention.n_heads, pk, next_q=nq)
            return r if nq is not None else (r, None)
        x = ops.linear(pooled.contiguous(), G)                   # pooled @ G^T -> [B, C]
        att = attention(query=self.label_query, key=x, value=x)  # [B, NLQ, 300]
        att = ops.linear(att, linear_5.weight.detach(), linear_5.bias.detach()).view(pooled.shape[0], -1)
        return ops.linear(att, x_linear.weight.detach(), x_linear.bias.detach()), None

    def _features(self, trunk, x):
        if x.dim() == 4 and x.shape[1] == 2048:
            return x              # precomputed trunk output
        return trunk(x)

    def forward(self, text, text_lens, text_mask, object_feature, place_feature, object_inp, place_inp,
                return_last_state=True):
        if self.training:
            raise RuntimeError("Multi_GCN_Multihead_Att: eval-mode forward only on the HIP path; call .eval()")
        if self.label_query is None:
            raise RuntimeError("label query missing: pass label_glove=... / opt['label_glove'], call "
                               "set_label_query(), or run from a directory holding %s" % (LABEL_GLOVE_CANDIDATES,))
        for name, t in (("text", text), ("object_feature", object_feature), ("place_feature", place_feature)):
            if not t.is_cuda:
                raise RuntimeError("%s is on %s: mgnns_amd operators run on the GPU only (no CPU path)" % (name, t.device))
        if self.label_query.device != text.device:
            self.label_query = self.label_query.to(text.device)
        with torch.no_grad():
            return self._forward_streams(text, text_lens, text_mask, object_feature, place_feature,
                                         object_inp, place_inp)

    def _side_streams(self, device):
        """The side streams of the forward's schedule by key ('s1'..'s3'), each on a hardware queue of its own
        (mgnns_amd.streams measures the binding once)."""
        key = str(device)      # probed against the stream that is current at FIRST use (never inside a graph capture)
        if self._streams is None or self._streams[0] != key:
            from .streams import independent_streams
            n = max(int(k[1:]) for sch in self.SCHEDULES.values() for _, k in sch if k != "main")
            chosen, distinct = independent_streams(device, n)
            self._streams = (key, {"s%d" % (i + 1): c for i, c in enumerate(chosen)}, distinct)
        return self._streams[1]

    # data dependencies between the forward's segments (MODEL:444-567)
    SEGMENT_DEPS = {
        "text_gcn": (), "text_bank": (), "lgcn_obj": (), "lgcn_place": (), "bank_obj": (), "bank_place": (),
        "tail_obj": ("lgcn_obj", "bank_obj"), "tail_place": ("lgcn_place", "bank_place"),
        "tio": ("bank_obj", "text_gcn"), "tip": ("bank_place", "text_gcn"),
        "iot": ("text_bank", "tail_obj"), "ipt": ("text_bank", "tail_place"),
        "head": ("tio", "tip", "iot", "ipt"),
    }
    # schedules: enqueue order and stream of every segment ('main' = the caller's stream, 's1'..'s3' side streams, all of
    # default priority on hardware queues of their own, mgnns_amd/streams.py).  Decided by bench.py measurements
    # (DESIGN.md section 6).
    SCHEDULES = {
        # one stream per channel, stacks where their producer ran (the round-1 schedule)
        # (the BiLSTM chain is the longest of the forward: it is enqueued FIRST -- at B=32 the single-graph runtime started
        #  it 199 us into a 587-us replay when it was captured sixth)
        "channels": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("lgcn_place", "s2"),
                     ("bank_place", "s2"), ("tail_obj", "s1"), ("tail_place", "s2"), ("tio", "main"), ("tip", "s3"),
                     ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        # the same, but a text->image stack also waits for its channel's label-attention tail: the tail's two small
        # launches then run BEFORE the chip-filling attention cores start instead of queueing for CUs behind them
        "tails_first": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("lgcn_place", "s2"), ("bank_place", "s2"),
                        ("tail_obj", "s1"), ("tail_place", "s2"), ("tio+tail_obj", "main"),
                        ("tip+tail_place", "s3"), ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        # memory banks first on their streams (they start at ~20 us instead of behind their label GCN); both label GCNs on
        # the text-GCN stream, which is otherwise idle until the place bank is done
        "banks_first": [("text_bank", "main"), ("bank_obj", "s1"), ("bank_place", "s2"), ("text_gcn", "s3"), ("lgcn_obj", "s3"),
                        ("lgcn_place", "s3"), ("tail_obj", "s1"), ("tail_place", "s2"), ("tio", "main"), ("tip", "s3"),
                        ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        # 'channels' with the object-side stacks re-homed: the text->object stack right behind the object bank (not behind the
        # BiLSTM chain on the caller's stream, which ends later than the bank at small batches), the image->text stack behind the
        # BiLSTM it needs, the object tail on the text-GCN stream
        "channels2": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("lgcn_place", "s2"),
                      ("bank_place", "s2"), ("tio", "s1"), ("tail_obj", "s3"), ("tail_place", "s2"), ("tip", "s3"),
                      ("iot", "main"), ("ipt", "s2"), ("head", "main")],
        # 'channels' with the two HBM-bound memory-bank kernels one after the other instead of side by side: together they
        # take as long either way, but the first one -- and the stack and the label tail behind it -- is done in half the time
        "banks_serial": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("lgcn_place", "s2"),
                         ("bank_place+bank_obj", "s2"), ("tail_obj", "s1"), ("tail_place", "s2"), ("tio", "main"), ("tip", "s3"),
                         ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        # 'channels' with the place channel's memory bank in FRONT of its label GCN (the longer one, C = 365: 105-123 us): the bank
        # starts at t = 0 instead of behind it
        "place_bank_first": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("bank_place", "s2"),
                             ("lgcn_place", "s2"), ("tail_obj", "s1"), ("tail_place", "s2"), ("tio", "main"), ("tip", "s3"),
                             ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        # ('place_bank_first' with the object bank WAITING for the place bank -- the two HBM-bound kernels one after the other,
        #  place first -- loses: 0.644-0.647 / 0.715-0.721 ms against 0.613-0.621 / 0.678-0.682, three alternating runs, round 5)
        # the same with the place label GCN on the text-GCN stream (idle until the place bank is done)
        "place_bank_first_lgcn_s3": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("bank_place", "s2"),
                                     ("lgcn_place", "s3"), ("tail_obj", "s1"), ("tail_place", "s2"), ("tio", "main"), ("tip", "s3"),
                                     ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        "tails_first_obj": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("lgcn_place", "s2"),
                            ("bank_place", "s2"), ("tail_obj", "s1"), ("tail_place", "s2"), ("tio+tail_obj", "main"), ("tip", "s3"),
                            ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        "tails_first_place": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("lgcn_place", "s2"),
                              ("bank_place", "s2"), ("tail_obj", "s1"), ("tail_place", "s2"), ("tio", "main"), ("tip+tail_place", "s3"),
                              ("iot", "s1"), ("ipt", "s2"), ("head", "main")],
        # chip-filling kernels in two chains, every small launch on a stream of its own:
        #   main: BiLSTM -> head;  s1: both image banks, then the two masked stacks;  s2: the two text->image stacks;
        #   s3: text GCN, label GCNs, label-attention tails
        # (round 5: a six-stream schedule -- BiLSTM, text GCN, two label GCNs and two memory banks each at the head of a stream of
        #  its own -- lost at every batch: 0.536 against 0.388 ms at B = 32 on the runtime's four hardware queues, where the extra
        #  streams share a queue, and GPU_MAX_HW_QUEUES=8 slows EVERY schedule down, 0.54-0.72 ms at B = 32 and 0.93 at B = 256:
        #  NOTES_r05 section 5)
        # small batches (round 5; default below 64 samples): the memory banks in FRONT of the two shorter heads -- object label
        # GCN, text GCN -- instead of behind the label GCNs ('channels2': label GCN 75 / 105 us, THEN the bank 59 / 54 us, then the
        # tail); the place label GCN (the longest head, C = 365) and the BiLSTM start at t = 0 on streams of their own; the
        # text->object stack behind the BiLSTM, the object->text stack behind the object tail.  B = 32: 0.305-0.325 ms two in
        # flight / 0.369-0.370 one at a time against 0.348-0.352 / 0.387-0.390 ('channels2', same box); B = 16: 0.310 / 0.361
        # against 0.340 / 0.378; B = 48: 0.330 / 0.396 against 0.357 / 0.396; at B = 64 it loses (0.42 against 0.37)
        "small": [("text_bank", "main"), ("lgcn_place", "s2"), ("bank_obj", "s1"), ("bank_place", "s3"), ("lgcn_obj", "s1"),
                  ("text_gcn", "s3"), ("tail_place", "s2"), ("tail_obj", "s1"), ("tip", "s3"), ("tio", "main"), ("ipt", "s2"),
                  ("iot", "s1"), ("head", "main")],
        # mid-size batches (found by tools/dev/sched_search.py from 'channels2' at B = 64; NOT the default): both memory banks behind
        # the object label GCN on ONE stream, the place label GCN (the longest head) alone on its own.  Throughput for latency: B = 64:
        # 0.348-0.352 ms two in flight / 0.456-0.457 one at a time against 0.369-0.370 / 0.418-0.419 ('channels2'); B = 96: 0.402-0.404 /
        # 0.486-0.492 against 0.425-0.439 / 0.470; B = 48: 0.310-0.317 / 0.453-0.455 against 0.326 / 0.397-0.405 ('small'); B = 128: loses
        "mid": [("text_bank", "main"), ("text_gcn", "s3"), ("lgcn_obj", "s1"), ("bank_obj", "s1"), ("lgcn_place", "s2"),
                ("bank_place", "s1"), ("tio", "s1"), ("tail_place", "s2"), ("tail_obj", "s3"), ("iot", "main"), ("tip", "s3"),
                ("ipt", "s2"), ("head", "main")],
        "bigsmall": [("text_gcn", "s3"), ("bank_obj", "s1"), ("text_bank", "main"), ("lgcn_obj", "s3"), ("bank_place", "s1"),
                     ("lgcn_place", "s3"), ("tio", "s2"), ("tail_obj", "s3"), ("tail_place", "s3"), ("tip", "s2"),
                     ("iot", "s1"), ("ipt", "s1"), ("head", "main")],
    }

    def resolve_schedule(self, batch, schedule=None):
        """Name of the schedule a forward of `batch` samples runs.  'auto': at 128 samples and more the chip-filling kernels
        decide: 'place_bank_first' -- 'channels' with the place channel's memory bank in FRONT of its label GCN (the longest head of a
        channel's chain, C = 365).  Round 3 measured no difference to 'channels'; with round 4's kernels (three alternating runs on
        one box, B = 256): 0.647-0.654 against 0.665-0.673 ms with two forwards in flight, 0.690-0.703 against 0.717-0.721 one at a
        time; equal at B = 128 (0.502 / 0.503).  Below 128 the BiLSTM chain on the caller's stream is the longest segment and
        'channels2' (no text->image stack queued behind it) wins: 0.381 vs 0.390 / 0.394 ms at B = 32, 0.412 vs 0.413 / 0.424 at
        B = 64 ('channels' / 'place_bank_first').  Below 64 (round 5): 'small' -- the memory banks in front of the shorter heads."""
        name = schedule or self.schedule
        if name == 'auto':
            # 'bf16x3' (three times the matrix work in the chip-filling kernels, the fp32 recurrence): 'channels2' at every batch --
            # B = 256: 1.499-1.505 ms against 1.534-1.537 two in flight, three alternating runs; B = 128: 0.86 against 0.99; B = 32:
            # 0.571 against 0.582-0.592 for 'small' (NOTES_r05 section 1)
            name = default_schedule(batch, getattr(self, 'precision', 'fp32'))
        if name not in self.SCHEDULES:
            raise ValueError("unknown schedule %r (one of %s, or 'auto')" % (name, sorted(self.SCHEDULES)))
        return name

    def forward_plan(self, text, text_lens, text_mask, object_feature, place_feature, object_inp, place_inp, schedule=None,
                     split_head=None):
        """The forward as SEGMENTS: a list of (name, stream key, names of the segments on OTHER streams it waits for,
        callable) in enqueue order.  The channels and the four fusion stacks are independent of each other
        (MODEL:444-546); each segment is a linear chain of launches on one stream and every cross-stream dependency is
        a wait on the event recorded right behind the producing segment -- never on a whole stream, which would also
        wait for whatever is queued behind the producer.  Eager execution (_forward_streams) and hipGraph capture
        (mgnns_amd.graph.GraphedForward: one linear graph per segment) both run this list; results go into the returned
        context dict (ctx['logits'] at the end)."""
        if not self.bidirectional:
            raise NotImplementedError("the HIP text bank implements the bidirectional LSTM the reference configures")
        ctx = _PlanCtx()
        fused_bf16 = (self.precision in ('bf16', 'bf16x3') and self.fused_label_tail and self.fused_label_tail_bf16
                      and text.shape[0] >= self.fused_label_tail_bf16_min_batch)
        plan_kind = mask_plan_applies(text_mask, self.precision, self.attention)        # False | 'packed' | 'grouped'

        def text_gcn():
            ops.stamp("text GCN start")
            tf = ctx['text_feature'] = self.text_features(text)
            # first projected queries of the two text->image stacks: here, early, instead of in front of their cores
            for nm, layers in (('tio', self.text_img_object_multi_head_att), ('tip', self.text_img_place_multi_head_att)):
                if len(layers):
                    ctx['qh_' + nm] = first_query(layers, tf)
            if plan_kind and PLAN_SITE == "text_gcn":
                # the mask's packing plan for both image->text stacks (MODEL:509-527), here: this stream is idle from now until
                # the place bank is done, the stacks that take the plan start ~200 us later (and wait for this segment)
                ctx['mha_plan'] = make_mask_plan(text_mask.float().contiguous(), self.precision, self.attention)
            ops.stamp("text GCN end")

        def text_bank():
            ops.stamp("main: start")
            # input conversion first: a bool / int mask (the reference documents a bool tensor) is cast HERE, in the
            # segment both masked stacks wait for
            ctx['text_mask'] = text_mask.float().contiguous()
            # the packing plan of the mask for both image->text stacks (MODEL:509-527) rides on the BiLSTM's prep launch: one more
            # workgroup there instead of a launch per channel (round 4) on the stacks' critical paths
            ctx['text_bank'] = self._text_bank(text, text_lens, ctx['text_mask'] if (PLAN_IN_PREP and plan_kind == 'packed') else None)
            # bf16x3 + faithful: BOTH masked stacks read the bank's split-bf16 (hi + lo) images and run on different streams --
            # the images are made HERE, in the segment both wait for (made lazily inside the first stack, the other stack's
            # core had no event ordering it behind the conversion launch: a stale read under graph replay)
            if any(m.slf_attn._split_core() for st in (self.img_object_text_multi_head_att, self.img_place_text_multi_head_att)
                   for m in st):
                ctx['text_bank_split'] = ctx['text_bank'].split      # (MemoryBank caches it: the stacks find the images made)
            prep_plan = getattr(ctx['text_bank'], 'mask_plan', None)
            if prep_plan is not None:
                ctx['mha_plan'] = prep_plan
            elif PLAN_IN_PREP and plan_kind:
                # (no prep launch to ride on -- the fp32 LSTM of bf16x3 mode, a shape the fused prep does not take: a launch here)
                ctx['mha_plan'] = make_mask_plan(ctx['text_mask'], self.precision, self.attention)
            ops.stamp("text bank (LSTM) end")

        def lgcn(tag, A, inp, attention):
            def run_fused():
                # gen_adj + both GraphConvolutions + the packed image of G + w_q(label query): ONE persistent launch
                pk = self._lgcn_pack(tag)
                # grid: a quarter of the CUs next to the chip-filling kernels of a large batch, 3/8 below (measured with the
                # item queue, ms per forward at grid 64 / 96 / 128: B=256 0.828 / 0.839 / 0.859, B=128 0.590 / 0.580 / 0.589,
                # B=32 0.448 / 0.433 / 0.433); any grid is correct
                n_cu = torch.cuda.get_device_properties(A.device).multi_processor_count
                grid = ops.LABEL_GCN_GRID or (n_cu // 4 if text.shape[0] >= 192 else 3 * n_cu // 8)
                G, Gp, Q = ops.label_gcn(A.detach(), inp[0].float().contiguous(), pk, want_packed_g=fused_bf16,
                                         query=(self.label_query.float().contiguous(), attention.w_q.weight.detach(),
                                                attention.w_q.bias.detach()), grid=grid)
                ctx['Q_' + tag], ctx['G_' + tag] = Q, G
                if fused_bf16:
                    ctx['Gp_' + tag] = Gp
                ops.stamp("  label GCN end")

            # the one-launch form has shape limits (C <= 512, widths multiples of 256, ...): everything else takes the
            # separate operators, which have none -- e.g. a 1000-class object model
            if self.fused_label_gcn and self._lgcn_fused_ok(A.shape[0], inp.shape[-1]):
                return run_fused

            def run():
                ctx['Q_' + tag] = self._label_q(attention)       # batch independent: off the critical path, with the GCN
                ctx['G_' + tag] = self._label_gcn(A, inp)
                if fused_bf16:                                   # its fragment-major bf16 image for the fused tail's read-out
                    ctx['Gp_' + tag] = ops.pack_weight_bf16_split(ctx['G_' + tag])
            return run

        def bank(tag, trunk, feature, lin):
            def run():
                ops.stamp("%s: bank start" % tag)
                feats = self._features(trunk, feature)
                setattr(self, 'object_feature' if tag == 'obj' else 'place_feature', feats)      # MODEL:450,482 keep them
                ctx['bank_' + tag], ctx['pooled_' + tag] = self._img_bank_and_pool(feats, lin)
                ops.stamp("%s: bank end" % tag)
            return run

        def tail(tag, attention, linear_5, x_linear, next_stack, next_name):
            def run():
                # the packing plan of the text mask for the image->text stack this tail feeds (which samples share a workgroup of
                # its masked attention launches, MODEL:509-527): one small launch per channel, HERE -- on the stack's own stream,
                # which has slack; on the BiLSTM's stream (the longest chain) it cost the pipelined forward 3 %, and one plan for
                # both stacks means a cross-stream dependency the runtime's one-graph capture of the schedule does not survive
                if plan_kind and PLAN_SITE == "tails":
                    ctx['mha_plan_' + next_name] = make_mask_plan(text_mask, self.precision, self.attention)
                ctx['att_' + tag], ctx['qh_' + next_name] = self._channel_tail(
                    ctx['pooled_' + tag], ctx['G_' + tag], ctx.get('Gp_' + tag), ctx['Q_' + tag], attention, linear_5, x_linear,
                    next_stack)
                ops.stamp("%s: label-attention tail end" % tag)
            return run

        # The classifier is linear in the four fusion features: with split_head every stack's chain ends with ITS share of the
        # logits (ops.classifier_head_part) and the last share to land completes them -- no head segment, i.e. no launch behind
        # events of three other streams at the very end of the forward (15-28 us in the timelines).  Not with a `post` step
        # captured behind the head (the sharded forward's all-gather): that needs the head as a segment.
        if split_head is None:
            split_head = self.split_head
        split_head = bool(split_head and self.fused_head)
        if split_head:
            # building the plan launches nothing: the executor calls ctx.prepare() on the caller's stream before it forks
            def prepare():
                wc, bc = self._head_pack()
                ctx['_head'] = (wc, bc, ops.classifier_head_state(text.shape[0], wc.shape[0], 4, wc.device))
                ctx['logits'] = ctx['_head'][2][2]
            ctx.prepare = prepare
        part_of = {"tio": 0, "tip": 1, "iot": 2, "ipt": 3}                # order of the reference's torch.cat (MODEL:560)

        def stack(name, layers, q_key, bank_key, masked):
            def run():
                ops.stamp("%s stack start" % name)
                ctx[name] = run_stack(layers, ctx[q_key], ctx[bank_key], ctx['text_mask'] if masked else None,
                                      qh=ctx.get('qh_' + name),
                                      plan=(ctx.get('mha_plan_' + name) if ctx.get('mha_plan_' + name) is not None
                                            else ctx.get('mha_plan')) if masked else None)
                if split_head:
                    wc, bc, hstate = ctx['_head']
                    ops.classifier_head_part(ctx[name], part_of[name], 4, wc, bc, hstate)
                ops.stamp("%s stack end" % name)
            return run

        def head():
            if self.fused_head:      # one launch: multi_linear_2 . multi_linear_1 composed (eval: nothing between them)
                wc, bc = self._head_pack()
                ctx['logits'] = ops.classifier_head([ctx['tio'], ctx['tip'], ctx['iot'], ctx['ipt']], wc, bc)
                ops.stamp("logits")
                return
            multi_feature = torch.cat([ctx['tio'], ctx['tip'], ctx['iot'], ctx['ipt']], dim=1)
            multi_feature = ops.linear(multi_feature, self.multi_linear_1.weight.detach(),
                                       self.multi_linear_1.bias.detach())
            ctx['logits'] = ops.linear(multi_feature, self.multi_linear_2.weight.detach(), self.multi_linear_2.bias.detach())
            ops.stamp("logits")

        fns = {
            "text_gcn": text_gcn, "text_bank": text_bank,
            "lgcn_obj": lgcn('obj', self.object_A, object_inp, self.object_attention),
            "lgcn_place": lgcn('place', self.place_A, place_inp, self.place_attention),
            "bank_obj": bank('obj', self.object_features, object_feature, self.liner_img_object),
            "bank_place": bank('place', self.place_features, place_feature, self.liner_img_place),
            "tail_obj": tail('obj', self.object_attention, self.object_linear_5, self.object_x_linear,
                             self.img_object_text_multi_head_att, 'iot'),
            "tail_place": tail('place', self.place_attention, self.place_linear_5, self.place_x_linear,
                               self.img_place_text_multi_head_att, 'ipt'),
            # text->image stacks need the image bank alone (not the channel's label-attention tail) and the text GCN
            "tio": stack("tio", self.text_img_object_multi_head_att, 'text_feature', 'bank_obj', False),
            "tip": stack("tip", self.text_img_place_multi_head_att, 'text_feature', 'bank_place', False),
            # image->text stacks need the text bank (and the mask cast next to it) and the channel's tail
            "iot": stack("iot", self.img_object_text_multi_head_att, 'att_obj', 'text_bank', True),
            "ipt": stack("ipt", self.img_place_text_multi_head_att, 'att_place', 'text_bank', True),
            "head": None if split_head else head,       # (None: nothing left to launch)
        }
        sched = self.SCHEDULES[self.resolve_schedule(text.shape[0], schedule)]
        where, plan = {}, []
        for entry, skey in sched:
            name, *extra = entry.split("+")          # "seg+other": also wait for `other` (ordering only, no data)
            deps = tuple(self.SEGMENT_DEPS[name]) + tuple(extra)
            if name in ("iot", "ipt") and plan_kind and PLAN_SITE == "text_gcn" and "text_gcn" not in deps:
                deps += ("text_gcn",)                # the mask's packing plan is built at the end of that segment
            for d in deps:
                if d not in where:
                    raise ValueError("schedule runs %s before %s" % (name, d))
            plan.append((name, skey, tuple(d for d in deps if where[d] != skey), fns[name]))
            where[name] = skey
        if set(where) != set(self.SEGMENT_DEPS):
            raise ValueError("schedule must run every segment exactly once")
        return plan, ctx

    def _forward_streams(self, *args):
        """Eager execution of forward_plan on four HIP streams (fork / join with events, no host sync)."""
        main = torch.cuda.current_stream()
        side = self._side_streams(args[0].device) if self.use_streams else {}
        streams = dict(side, main=main)
        for k in {k for _, k in self.SCHEDULES[self.resolve_schedule(args[0].shape[0])]}:
            streams.setdefault(k, main)              # use_streams = False: everything on the caller's stream
        plan, ctx = self.forward_plan(*args)
        ctx.prepare()                                # (the split head's zeroed counter: on `main`, before the fork)
        for st in side.values():
            st.wait_stream(main)                     # the caller produced the inputs on `main`
        needed = {d for _, _, deps, _ in plan for d in deps}
        done = {}
        for name, skey, deps, fn in plan:
            if fn is None:
                continue
            st = streams[skey]
            for d in deps:
                if done[d][1] is not st:
                    st.wait_event(done[d][0])
            with torch.cuda.stream(st):
                fn()
                if name in needed:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    done[name] = (ev, st)
        for st in side.values():
            main.wait_stream(st)                     # also orders every side-stream allocation before the caller's reuse
        return ctx['logits']

    def get_config_optim(self, lr, lrp):
        return [
            {'params': self.text_features.parameters(), 'lr': lr * 10},
            {'params': self.object_features.parameters(), 'lr': lr * lrp},
            {'params': self.place_features.parameters(), 'lr': lr * lrp},
            {'params': self.gc1.parameters(), 'lr': lr},
            {'params': self.gc2.parameters(), 'lr': lr},
            {'params': self.object_attention.parameters(), 'lr': lr},
            {'params': self.place_attention.parameters(), 'lr': lr},
            {'params': self.lstm.parameters(), 'lr': lr * 10},
            {'params': self.img_object_text_multi_head_att.parameters(), 'lr': lr},
            {'params': self.img_place_text_multi_head_att.parameters(), 'lr': lr},
            {'params': self.text_img_object_multi_head_att.parameters(), 'lr': lr},
            {'params': self.text_img_place_multi_head_att.parameters(), 'lr': lr},
        ]


def Text_model_from_parts(vocab, edges_mappings, count, num_labels, ngram, text_dropout, edges_weights=None):
    """Text channel from an already built vocabulary, edge map (dense, scipy sparse or PmiCsr) and edge count
    (what Text_model below derives from the training split)."""
    return Text_GCN_Model(num_labels, hidden_size_node=300, vocab=vocab, n_gram=ngram, drop_out=text_dropout,
                          edges_matrix=edges_mappings, edges_num=count, pmi=edges_weights, cuda=True,
                          trainable_edges=True)


def get_content(data_root_path):
    """The `text` field of every line of <data_root>/all_anno_json/train_all_anno.json (utils/pmi.py:17-25)."""
    import json
    all_text = []
    with open(os.path.join(data_root_path, 'all_anno_json', 'train_all_anno.json'), 'r') as f:
        for line in f:
            all_text.append(json.loads(line)['text'])
    return all_text


def Text_model(data_root_path, vocab_root_path, text_min_count, window_size,
               num_labels, ngram, text_dropout, min_cooccurence):
    """MODEL:598-615 with the reference's signature: vocabulary from <vocab_root>/vocab/vocab-<N>.txt (built from the
    train split when absent, utils/vocab_new.py:8-15), PMI edge map from the train split (utils/pmi.py:28-105 -- here the
    sparse builder of mgnns_amd.pmi: same ids, same weights, no dense [V,V] matrices), Text_GCN.Model on top."""
    from .pmi import build_pmi
    from .vocab import get_vocab_list
    vocab = get_vocab_list(data_root_path, vocab_root_path, text_min_count)
    edges_weights, edges_mappings, count = build_pmi(get_content(data_root_path), vocab, window_size=window_size,
                                                     min_cooccurence=min_cooccurence)
    return Text_GCN_Model(num_labels, hidden_size_node=300, vocab=vocab, n_gram=ngram, drop_out=text_dropout,
                          edges_matrix=edges_mappings, edges_num=count, pmi=torch.from_numpy(edges_weights),
                          cuda=True, trainable_edges=True)


def _weights_dir():
    return os.environ.get('MGNNS_WEIGHTS_DIR', 'weights')


def _trunk_init_random():
    """MGNNS_TRUNK_INIT=random: build the trunks without a checkpoint (their weights then arrive through
    load_state_dict, e.g. a resumed MGNNS checkpoint -- ENGINE:399,415).  Default: a missing checkpoint raises."""
    return os.environ.get('MGNNS_TRUNK_INIT', '') == 'random'


def _load_checkpoint(path):
    """torch.load with the safe unpickler.  Checkpoints that pickle numpy scalars next to the tensors (Places365 ships
    'best_prec1' as one) get exactly those globals allow-listed; a file that still needs the full unpickler -- arbitrary code
    execution by design -- is only loaded on the explicit opt-in MGNNS_TRUST_CHECKPOINTS=1."""
    import pickle
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as first:
        try:
            import numpy as np
            safe = [np.core.multiarray.scalar, np.dtype] if hasattr(np, "core") else []
            try:
                from numpy import dtypes as _npd
                safe += [getattr(_npd, n) for n in dir(_npd) if n.endswith("DType")]
            except ImportError:
                pass
            with torch.serialization.safe_globals(safe):
                return torch.load(path, map_location='cpu', weights_only=True)
        except (pickle.UnpicklingError, AttributeError):
            pass
        if os.environ.get('MGNNS_TRUST_CHECKPOINTS', '') == '1':
            import warnings
            warnings.warn("loading %s with the full (code-executing) unpickler: MGNNS_TRUST_CHECKPOINTS=1" % path)
            return torch.load(path, map_location='cpu', weights_only=False)
        raise RuntimeError("%s needs the full unpickler (%s); set MGNNS_TRUST_CHECKPOINTS=1 if you trust the file" % (path, first))


def place_resnet(arch='resnet50'):
    """MODEL:586-595: the Places365 ResNet from weights/<arch>_places365.pth.tar ('module.' prefixes stripped)."""
    from . import trunk
    if arch != 'resnet50':
        raise ValueError("the HIP trunk implements the bottleneck ResNets the reference uses; got arch=%r" % (arch,))
    model = trunk.resnet50(num_classes=365)
    model_file = os.path.join(_weights_dir(), '%s_places365.pth.tar' % arch)
    if not os.path.exists(model_file):
        if _trunk_init_random():
            return model
        raise FileNotFoundError("%s not found (set MGNNS_WEIGHTS_DIR, or MGNNS_TRUNK_INIT=random to build the trunk "
                                "uninitialised and load a full model checkpoint afterwards)" % model_file)
    checkpoint = _load_checkpoint(model_file)
    state_dict = {str.replace(k, 'module.', ''): v for k, v in checkpoint['state_dict'].items()}
    model.load_state_dict(state_dict)
    return model


def object_resnet(pretrained=True):
    """MODEL:629 `models.resnet101(pretrained=pretrained)`.  There is no download here: pretrained=True reads the
    torchvision checkpoint (resnet101-*.pth) from MGNNS_WEIGHTS_DIR / weights/ or torch's hub cache."""
    import glob
    from . import trunk
    model = trunk.resnet101()
    if not pretrained:
        return model
    hub = os.path.join(os.environ.get('TORCH_HOME', os.path.expanduser('~/.cache/torch')), 'hub', 'checkpoints')
    found = []
    for d in (_weights_dir(), hub):
        found += sorted(glob.glob(os.path.join(d, 'resnet101*.pth')))
    if not found:
        if _trunk_init_random():
            return model
        raise FileNotFoundError("pretrained=True but no resnet101*.pth under %s or %s (no network to download it; set "
                                "MGNNS_WEIGHTS_DIR, pass pretrained=False, or MGNNS_TRUNK_INIT=random)" % (_weights_dir(), hub))
    model.load_state_dict(_load_checkpoint(found[0]))
    return model


def multi_gcn_multihead_att_model(opt,
                                  num_labels,
                                  object_num_classes, place_num_classes, object_t, place_t,
                                  data_root_path, vocab_root_path,
                                  text_min_count, window_size,
                                  ngram, min_cooccurence,
                                  text_dropout=0.5,
                                  pretrained=True,
                                  object_adj_file=None, place_adj_file=None, in_channel=300):
    """MODEL:619-642, same parameters in the same order (the keyword call of Tumblr_Multi_GCN_Multihead_Att.py:144-157
    works unchanged): ResNet-101 + Places365 ResNet-50 trunks on the HIP convolution kernels, the text channel built
    from the train split, Multi_GCN_Multihead_Att on top."""
    object_model = object_resnet(pretrained=pretrained)
    place_model = place_resnet()
    text_model = Text_model(data_root_path, vocab_root_path, text_min_count, window_size,
                            num_labels, ngram, text_dropout, min_cooccurence)
    return Multi_GCN_Multihead_Att(opt, num_labels, text_model=text_model,
                                   object_model=object_model, place_model=place_model,
                                   object_num_classes=object_num_classes, place_num_classes=place_num_classes,
                                   in_channel=in_channel, object_t=object_t, place_t=place_t,
                                   object_adj_file=object_adj_file, place_adj_file=place_adj_file)


def multi_gcn_multihead_att_model_from_parts(opt, num_labels, object_num_classes, place_num_classes, object_t, place_t,
                                             text_model, object_model=None, place_model=None,
                                             object_adj_file=None, place_adj_file=None, in_channel=300, label_glove=None):
    """The same module with the text model and the (optional) CNN trunks injected instead of being built from the
    data root (synthetic benchmarks, tests; trunks None = feature-map entry only)."""
    return Multi_GCN_Multihead_Att(opt, num_labels, text_model=text_model, object_model=object_model,
                                   place_model=place_model, object_num_classes=object_num_classes,
                                   place_num_classes=place_num_classes, in_channel=in_channel,
                                   object_t=object_t, place_t=place_t, object_adj_file=object_adj_file,
                                   place_adj_file=place_adj_file, label_glove=label_glove)
