"""Operator wrappers: torch CUDA tensors in, torch CUDA tensors out, all arithmetic in
libmgnns_hip.so on the caller's current stream.  Shape / dtype / device / contiguity are
validated here (the reference's convention is Python exceptions); nothing in this file
computes.
"""
import os

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU2 = 0, 1, 2
BANK_LD = 320      # bf16 memory banks are [B, L, 320]: model dim 300 zero padded to 10 MFMA k-steps of 32


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Scratch buffers of the persistent launches are keyed by (capture epoch, launch stream): eager launches on one stream
# are ordered and may share a scratch; every hipGraph capture takes a fresh epoch (mgnns_amd/graph.py), because a captured
# graph can be replayed on any stream next to any other graph of the same model.
_SCRATCH_EPOCH = 0


def new_scratch_epoch():
    global _SCRATCH_EPOCH
    _SCRATCH_EPOCH += 1
    return _SCRATCH_EPOCH


def set_scratch_epoch(e):
    global _SCRATCH_EPOCH
    prev, _SCRATCH_EPOCH = _SCRATCH_EPOCH, int(e)
    return prev


def _scratch_key():
    return (_SCRATCH_EPOCH, _stream())


# Every scratch slot created under a capture epoch is remembered, so that the capture's owner (graph.GraphedForward) can drop
# them when it goes: slot dicts would otherwise pin a slice of the zero pool per (capture, launch) for the life of the process.
_EPOCH_SLOTS = {}


def _scratch_slot_put(slot, key, ws):
    slot[key] = ws
    if key[0]:
        _EPOCH_SLOTS.setdefault(key[0], []).append((slot, key))


def release_scratch_epoch(epoch):
    """Forget the scratch buffers created under `epoch` (its graphs are gone: nothing holds their addresses any more)."""
    for slot, key in _EPOCH_SLOTS.pop(int(epoch), []):
        slot.pop(key, None)


# Derived weight packs / scratch that a captured hipGraph may hold by raw address: while any capture is alive, a superseded
# pack is parked here instead of being freed (load_state_dict / set_precision with a live GraphedForward must not let a replay
# read recycled memory); the park empties when the last capture goes.  model._cache_put keeps its own per-model list; the
# layer-level caches of fusion.py (no back-pointer to the model) use this one.
_LIVE_CAPTURES = 0
_RETIRED = []


def capture_born():
    global _LIVE_CAPTURES
    _LIVE_CAPTURES += 1


def capture_gone():
    global _LIVE_CAPTURES
    if _LIVE_CAPTURES > 0:
        _LIVE_CAPTURES -= 1
    if _LIVE_CAPTURES == 0:
        _RETIRED.clear()


def retire(obj):
    """Park a superseded cache entry while a capture that may hold its addresses is alive."""
    if obj is not None and _LIVE_CAPTURES > 0:
        _RETIRED.append(obj)


# Counters of the persistent / cluster launches start at zero and are left at zero by the kernels themselves (the last
# workgroup re-arms them).  They come out of a pool that is zeroed OUTSIDE any capture: a torch.zeros inside a capture is a
# fill node of the graph -- a 5-8 us launch in front of the kernel on every replay (13 of them per forward, most on the
# bank -> tail -> stack chains).  Slices are handed out once and never reused.
_ZERO_POOL = {}
_ZERO_CHUNK = 64 << 20


def _capturing():
    return torch.cuda.is_current_stream_capturing()


def reserve_zeros(device, nbytes=24 << 20):
    """Make sure the zero pool of `device` has nbytes left (call outside a capture, e.g. right before one)."""
    device = torch.device(device)
    pool = _ZERO_POOL.get(device)
    if (pool is None or pool[1] + nbytes > pool[0].numel()) and not _capturing():
        buf = torch.zeros(max(_ZERO_CHUNK, nbytes), dtype=torch.uint8, device=device)
        torch.cuda.current_stream(device).synchronize()           # zero before any stream uses a slice
        _ZERO_POOL[device] = [buf, 0]


def zeros_bytes(nbytes, device):
    """nbytes of zeroed device memory (256-B aligned) that the caller's kernels keep zero between launches."""
    device = torch.device(device)
    n = (int(nbytes) + 255) // 256 * 256
    pool = _ZERO_POOL.get(device)
    if pool is None or pool[1] + n > pool[0].numel():
        if _capturing():                                          # pool exhausted inside a capture: a fill node, still correct
            return torch.zeros(n, dtype=torch.uint8, device=device)
        reserve_zeros(device, max(n, 24 << 20))
        pool = _ZERO_POOL[device]
    t = pool[0][pool[1]:pool[1] + n]
    pool[1] += n
    return t


def zeros_i32(n, device):
    return zeros_bytes(4 * n, device).view(torch.int32)[:n]


class KernelTimer:
    """Opt-in per-launch HIP-event timing (bench.py's roofline leg).  Events are recorded on the stream
    the kernels are launched on (PyTorch's current stream), around the C-ABI call only."""

    def __init__(self, names=None):
        self.names = None if names is None else set(names)
        self.events = {}

    def wants(self, name):
        return self.names is None or name in self.names

    def durations_ms(self):
        """{key: [ms, ...]} -- call after torch.cuda.synchronize()."""
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in self.events.items()}


_TIMER = None


def set_timer(timer):
    global _TIMER
    _TIMER = timer


def _launch(name, key, fn, *args):
    """Call one C-ABI entry point, optionally bracketed by events; raise on a non-zero return."""
    t = _TIMER
    if t is not None and t.wants(name):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        t.events.setdefault(key, []).append((a, b))
    else:
        rc = fn(*args)
    _lib.check(rc, name)


def _chk(t, name, dtype=torch.float32, ndim=None):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: mgnns_amd operators run on the GPU only (no CPU path)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must be %d-d, got shape %s" % (name, ndim, tuple(t.shape)))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _p(t):
    return None if t is None else t.data_ptr()


_gemm_ws = {}


def _gemm_workspace(device):
    """K-split scratch of the GEMM kernels: one buffer per (device, stream) so concurrent streams never share."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    buf = _gemm_ws.get(key)
    if buf is None:
        buf = torch.empty(_lib.lib().mgnns_gemm_workspace_bytes(), dtype=torch.uint8, device=device)
        _gemm_ws[key] = buf
    return buf


# ---- nn.Linear / matmul ---------------------------------------------------------------
def linear(x, weight, bias=None, act=ACT_NONE, residual=None):
    """act(x @ weight.T + bias) (+ residual); x [..., K], weight [N, K]."""
    K = x.shape[-1]
    x2 = _chk(x.reshape(-1, K), "x")
    _chk(weight, "weight", ndim=2)
    if weight.shape[1] != K:
        raise ValueError("weight %s does not match x[..., %d]" % (tuple(weight.shape), K))
    N = weight.shape[0]
    if bias is not None:
        _chk(bias, "bias", ndim=1)
    y = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = _chk(residual.reshape(-1, N), "residual")
        if residual.shape[0] != x2.shape[0]:
            raise ValueError("residual rows mismatch")
    L = _lib.lib()
    ws = _gemm_workspace(x.device)
    _lib.check(L.mgnns_linear_fwd(_p(x2), x2.shape[0], K, _p(weight), _p(bias), N, _p(residual), _p(y), act,
                                  _p(ws), ws.numel(), _stream()), "mgnns_linear_fwd")
    return y.view(*x.shape[:-1], N)


def classifier_head(feats, weight, bias):
    """logits = cat(feats, dim=1) @ weight.T + bias for the four fusion features [B, D] (MODEL:560-566), one launch, no cat."""
    if len(feats) != 4:
        raise ValueError("classifier_head takes the four fusion features")
    B, D = feats[0].shape
    for i, f in enumerate(feats):
        _chk(f, "feature %d" % i, ndim=2)
        if tuple(f.shape) != (B, D):
            raise ValueError("feature %d shape %s, expected %s" % (i, tuple(f.shape), (B, D)))
    _chk(weight, "weight", ndim=2)
    _chk(bias, "bias", ndim=1)
    NL = weight.shape[0]
    if weight.shape[1] != 4 * D or bias.shape[0] != NL:
        raise ValueError("weight %s / bias %s do not match four [B,%d] features" % (tuple(weight.shape), tuple(bias.shape), D))
    out = torch.empty(B, NL, device=weight.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.mgnns_classifier_head_fwd(_p(feats[0]), _p(feats[1]), _p(feats[2]), _p(feats[3]), B, D, _p(weight), _p(bias), NL,
                                           _p(out), _stream()), "mgnns_classifier_head_fwd")
    return out


def classifier_head_state(B, NL, nparts, device):
    """Buffers of one forward's split classifier head (classifier_head_part): parts [nparts,B,NL], arrival counter (zero),
    logits [B,NL]."""
    return (torch.empty(nparts, B, NL, device=device, dtype=torch.float32), zeros_i32(1, device),
            torch.empty(B, NL, device=device, dtype=torch.float32))


def classifier_head_part(feat, part, nparts, weight, bias, state):
    """This feature's share of logits = cat(feats) @ weight.T + bias (mgnns_classifier_part_fwd): call once per feature, from
    whichever stream produced it; the call that finishes last completes state[2] (the logits)."""
    _chk(feat, "feature", ndim=2)
    _chk(weight, "weight", ndim=2)
    _chk(bias, "bias", ndim=1)
    B, D = feat.shape
    parts, counter, logits = state
    NL = weight.shape[0]
    if weight.shape[1] != nparts * D or tuple(parts.shape) != (nparts, B, NL) or tuple(logits.shape) != (B, NL):
        raise ValueError("classifier_head_part: weight %s / state do not match %d features [B=%d,%d]" % (tuple(weight.shape), nparts, B, D))
    L = _lib.lib()
    _lib.check(L.mgnns_classifier_part_fwd(_p(feat), int(part), int(nparts), B, D, _p(weight), _p(bias), NL, _p(parts), _p(counter),
                                           _p(logits), _stream()), "mgnns_classifier_part_fwd")
    return logits


def matmul(x, w, act=ACT_NONE):
    """act(x @ w); x [M, K], w [K, N] (GraphConvolution weight layout)."""
    _chk(x, "x", ndim=2)
    _chk(w, "w", ndim=2)
    if x.shape[1] != w.shape[0]:
        raise ValueError("matmul shapes %s x %s" % (tuple(x.shape), tuple(w.shape)))
    y = torch.empty(x.shape[0], w.shape[1], device=x.device, dtype=torch.float32)
    L = _lib.lib()
    ws = _gemm_workspace(x.device)
    _lib.check(L.mgnns_matmul_fwd(_p(x), x.shape[0], x.shape[1], _p(w), w.shape[1], _p(y), act, _p(ws), ws.numel(),
                                  _stream()), "mgnns_matmul_fwd")
    return y


# ---- adjacency ----------------------------------------------------------------------------
def gen_adj(A, want_csr=False):
    """D^-1/2 A^T D^-1/2 (utils/util.py:421-426).  Returns adj, or (adj, (row_ptr, col, val))."""
    _chk(A, "A", ndim=2)
    C = A.shape[0]
    if A.shape[1] != C:
        raise ValueError("A must be square")
    adj = torch.empty_like(A)
    work = torch.empty(C, device=A.device, dtype=torch.float32)
    rp = col = val = None
    if want_csr:
        rp = torch.empty(C + 1, device=A.device, dtype=torch.int32)
        col = torch.empty(C * C, device=A.device, dtype=torch.int32)
        val = torch.empty(C * C, device=A.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.mgnns_gen_adj(_p(A), C, _p(adj), _p(work), _p(rp), _p(col), _p(val), _stream()), "mgnns_gen_adj")
    return (adj, (rp, col, val)) if want_csr else adj


def label_gcn_pack(w1, w2, split):
    """GraphConvolution weights W1 [K0,N1], W2 [N1,N2] ([in,out] layout) -> the packed dict label_gcn() takes:
    split = False: exact-fp32 fragment-major buffers; True: (hi, lo) split-bf16 pairs."""
    _chk(w1, "gc1.weight", ndim=2)
    _chk(w2, "gc2.weight", ndim=2)
    if w1.shape[1] != w2.shape[0]:
        raise ValueError("gc1.weight %s / gc2.weight %s do not chain" % (tuple(w1.shape), tuple(w2.shape)))
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()               # [N, K]: the layout the pack kernels take
    if split:
        a, b = pack_weight_bf16_split(w1t), pack_weight_bf16_split(w2t)
        d = {"w1": a, "w2": b}
    else:
        d = {"w1": (pack_weight_f32(w1t), None), "w2": (pack_weight_f32(w2t), None)}
    d.update(split=bool(split), K0=w1.shape[0], N1=w1.shape[1], N2=w2.shape[1])
    return d


LABEL_GCN_GRID = int(os.environ.get("MGNNS_LGCN_GRID", "0"))


def label_gcn(A, inp, packed, want_packed_g=False, query=None, grid=0):
    """One channel's whole label GCN in one persistent launch (mgnns_label_gcn_fwd): gen_adj + GraphConvolution x 2 (+ the
    label query projection).  A [C,C]; inp [C,K0]; packed = label_gcn_pack(...) (also carries this channel's scratch);
    query = (label_query [NLQ,K0], w_q.weight [HQ,K0], w_q.bias or None).  -> (G [C,N2], (Gp_hi, Gp_lo) or None, Q or None)."""
    _chk(A, "A", ndim=2)
    _chk(inp, "inp", ndim=2)
    C = A.shape[0]
    if A.shape[1] != C or inp.shape[0] != C or inp.shape[1] != packed["K0"]:
        raise ValueError("A %s / inp %s do not match a [C,C] adjacency and [C,%d] label embeddings" %
                         (tuple(A.shape), tuple(inp.shape), packed["K0"]))
    N1, N2 = packed["N1"], packed["N2"]
    L = _lib.lib()
    need = L.mgnns_label_gcn_scratch_bytes(C, N1, N2)
    # the launch's scratch holds its intermediates and its item queue: one per (capture epoch, launch stream), so that two
    # forwards of one model in flight at once -- two streams, two captured graphs -- never share one
    slot = packed.setdefault("_scratch", {})
    key = _scratch_key()
    ws = slot.get(key)
    if ws is None or ws.numel() < need or ws.device != A.device:
        if ws is not None:
            packed.setdefault("_retired", []).append(ws)     # a captured hipGraph may still hold its address: never freed
        ws = zeros_bytes(need, A.device)[:need]                           # counters (first 256 B) start at zero
        _scratch_slot_put(slot, key, ws)
    G = torch.empty(C, N2, device=A.device, dtype=torch.float32)
    gh = gl = None
    if want_packed_g:
        n = L.mgnns_packed_bf16_weight_bytes(C, N2)
        gh = torch.empty(n, dtype=torch.uint8, device=A.device)
        gl = torch.empty(n, dtype=torch.uint8, device=A.device)
    lq = wq = bq = Q = None
    nlq = hq = 0
    if query is not None:
        lq, wq, bq = query
        _chk(lq, "label_query", ndim=2)
        _chk(wq, "w_q.weight", ndim=2)
        if lq.shape[1] != packed["K0"] or wq.shape[1] != packed["K0"]:
            raise ValueError("label_query %s / w_q.weight %s must be [*, %d]" % (tuple(lq.shape), tuple(wq.shape), packed["K0"]))
        if bq is not None:
            _chk(bq, "w_q.bias", ndim=1)
        nlq, hq = lq.shape[0], wq.shape[0]
        Q = torch.empty(nlq, hq, device=A.device, dtype=torch.float32)
    _launch("mgnns_label_gcn_fwd", ("mgnns_label_gcn_fwd", C, packed["split"]), L.mgnns_label_gcn_fwd, _p(A), C, _p(inp), packed["K0"],
            1 if packed["split"] else 0, _p(packed["w1"][0]), _p(packed["w1"][1]), N1, _p(packed["w2"][0]), _p(packed["w2"][1]), N2,
            _p(G), _p(gh), _p(gl), _p(lq), nlq, _p(wq), _p(bq), hq, _p(Q), _p(ws), ws.numel(), int(grid) or LABEL_GCN_GRID, _stream())
    return G, ((gh, gl) if want_packed_g else None), Q


def dense_to_csr(m):
    _chk(m, "adj", ndim=2)
    C = m.shape[0]
    if m.shape[1] != C:
        raise ValueError("adj must be square")
    rp = torch.empty(C + 1, device=m.device, dtype=torch.int32)
    col = torch.empty(C * C, device=m.device, dtype=torch.int32)
    val = torch.empty(C * C, device=m.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.mgnns_dense_to_csr(_p(m), C, _p(rp), _p(col), _p(val), _stream()), "mgnns_dense_to_csr")
    return rp, col, val


def spmm_csr(csr, x, act=ACT_NONE, out=None, bias=None):
    """act(adj @ x [+ bias]) with adj in CSR (row_ptr, col, val); x [C, F].  out: optional preallocated [C, F] result;
    bias: GraphConvolution(bias=True)'s [1,1,F] parameter (MODEL:40-41,55-56), any shape with F elements."""
    rp, col, val = csr
    _chk(rp, "row_ptr", torch.int32, 1)
    _chk(col, "col", torch.int32, 1)
    _chk(val, "val", torch.float32, 1)
    _chk(x, "x", ndim=2)
    n = rp.shape[0] - 1
    if out is None:
        y = torch.empty(n, x.shape[1], device=x.device, dtype=torch.float32)
    else:
        y = _chk(out, "out", ndim=2)
        if tuple(y.shape) != (n, x.shape[1]) or y.data_ptr() == x.data_ptr():
            raise ValueError("out must be a [%d, %d] tensor distinct from x" % (n, x.shape[1]))
    L = _lib.lib()
    if bias is not None:
        bias = _chk(bias.reshape(-1), "bias", ndim=1)
        if bias.shape[0] != x.shape[1]:
            raise ValueError("bias has %d elements, x has %d features" % (bias.shape[0], x.shape[1]))
        _launch("mgnns_spmm_csr_fwd", ("mgnns_spmm_csr_bias_fwd", n, x.shape[1]), L.mgnns_spmm_csr_bias_fwd, _p(rp), _p(col), _p(val), n,
                _p(x), x.shape[1], _p(bias), _p(y), act, _stream())
        return y
    _launch("mgnns_spmm_csr_fwd", ("mgnns_spmm_csr_fwd", n, x.shape[1]), L.mgnns_spmm_csr_fwd, _p(rp), _p(col), _p(val), n,
            _p(x), x.shape[1], _p(y), act, _stream())
    return y


# ---- bf16-feature sparse propagation (BASELINE configs[4]) -------------------------------------------------
def cast_bf16(x):
    """fp32 tensor -> bf16 tensor of the same shape (round to nearest even), on the device."""
    _chk(x, "x")
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    if x.numel() == 0:
        return y
    _lib.check(_lib.lib().mgnns_cast_bf16(_p(x), x.numel(), _p(y), _stream()), "mgnns_cast_bf16")
    return y


class SparseAdjBf16:
    """A static sparse adjacency for `spmm_bf16`: CSR with bf16 values on the device, plus -- for graphs with tens of
    non-zeros per row -- the re-ordered entry stream of the LDS-tiled kernel (mgnns_amd/spmm_plan.py), built once per
    feature width on first use.  csr = (row_ptr int32 [n+1], col int32 [nnz], val fp32 or bf16 [nnz]) device tensors."""

    TILED_MIN_AVG_NNZ = 32.0

    def __init__(self, csr, n_cols=None, sort_rows=True):
        """sort_rows: hand the gather kernels the rows SORTED BY LENGTH (+ the map back to the rows of Y): four rows share a
        wave instruction there, and with Poisson row lengths the longest of four sets the pace (configs[4] at density 4e-4:
        21 -> 16 us).  One-off host work on a static adjacency; every row of Y is computed exactly as before."""
        rp, col, val = csr
        _chk(rp, "row_ptr", torch.int32, 1)
        _chk(col, "col", torch.int32, 1)
        if val.dtype == torch.float32:
            val = cast_bf16(_chk(val, "val", torch.float32, 1))
        _chk(val, "val", torch.bfloat16, 1)
        if val.numel() % 2:                                  # the kernels fetch the values as aligned dwords
            val = torch.cat([val, val.new_zeros(1)])[:-1]    # same values, storage readable one element further
        self.row_ptr, self.col, self.val = rp, col, val
        self.n_rows = rp.shape[0] - 1
        self.n_cols = self.n_rows if n_cols is None else int(n_cols)
        self.nnz = int(col.shape[0])
        self._plans = {}
        self._sort_rows = bool(sort_rows) and self.n_rows > 1
        self._sorted = {}                                    # form -> (row_ptr, col, val, row_map) with rows in order of length

    def sorted_for(self, F):
        """(row_ptr, col, val, row_map) for the gather kernel mgnns_spmm_csr_bf16_fwd picks at feature width F, or None
        (sort_rows=False, or all rows equally long).  Rows go in order of LENGTH -- for the ring form (one slab per XCD:
        a wave owns a CONTIGUOUS range of CSR positions) in quads of equal length dealt out with a stride, because sorted end
        to end the last waves would own all the long rows (measured: 33 us instead of 21); the register form deals rows out
        round robin and takes the plain sort."""
        if not self._sort_rows:
            return None
        form = "reg" if ((F + 127) // 128 + 7) // 8 >= 2 else "ring"      # (variant 0 of mgnns_spmm_csr_bf16_fwd)
        if form not in self._sorted:
            import numpy as np
            rph = self.row_ptr.cpu().numpy().astype(np.int64)
            lens = rph[1:] - rph[:-1]
            hit = None
            if lens.min() != lens.max():
                order = np.argsort(lens, kind="stable")                       # CSR position -> row of Y
                nq = self.n_rows // 4
                if form == "ring" and nq > 8:
                    stride = next(p for p in (389, 397, 401, 409, 419, 421, 431, 433) if nq % p)
                    q = (np.arange(nq, dtype=np.int64) * stride) % nq
                    order = np.concatenate([order[:4 * nq].reshape(nq, 4)[q].reshape(-1), order[4 * nq:]])
                rp2 = np.zeros(self.n_rows + 1, np.int64)
                np.cumsum(lens[order], out=rp2[1:])
                src = np.repeat(rph[:-1][order] - rp2[:-1], lens[order]) + np.arange(self.nnz)      # entry k of the sorted CSR <- entry src[k]
                idx = torch.from_numpy(src).to(self.col.device)
                val2 = self.val[:self.nnz][idx]
                if self.nnz % 2:
                    val2 = torch.cat([val2, val2.new_zeros(1)])[:-1]
                dev = self.row_ptr.device
                hit = (torch.from_numpy(rp2.astype(np.int32)).to(dev), self.col[idx].contiguous(), val2.contiguous(),
                       torch.from_numpy(order.astype(np.int32)).to(dev))
            self._sorted[form] = hit
        return self._sorted[form]

    @classmethod
    def block_diagonal(cls, adjs, **kw):
        """The union of independent graphs as ONE adjacency (rows / columns of graph i shifted by the sizes of graphs 0..i-1):
        act(A_i @ X_i) for every i = one launch on the stacked X -- configs[4]'s three channels (MODEL:460-506 runs the object
        and the scene graph one after the other): the fixed cost of a launch is paid once, 21.8 -> 17.4 us per channel."""
        rps, cols, vals, r0, c0, e0 = [], [], [], 0, 0, 0
        for a in adjs:
            rps.append(a.row_ptr[(1 if rps else 0):] + e0)
            cols.append(a.col + c0)
            vals.append(a.val[:a.nnz])
            r0, c0, e0 = r0 + a.n_rows, c0 + a.n_cols, e0 + a.nnz
        return cls((torch.cat(rps).to(torch.int32), torch.cat(cols).to(torch.int32), torch.cat(vals)), n_cols=c0, **kw)

    @property
    def avg_nnz(self):
        return self.nnz / max(1, self.n_rows)

    def plan(self, F, geometry=None):
        """Device copy of the tiled plan for feature width F (built on the host once; the adjacency is static)."""
        from . import spmm_plan
        geo = tuple(geometry) if geometry is not None else spmm_plan.geometry_for(self.n_rows, F)
        hit = self._plans.get(geo)
        if hit is None:
            bits = self.val.view(torch.int16).cpu().numpy().view("uint16")
            pl = spmm_plan.build_tiled_plan(self.row_ptr.cpu().numpy(), self.col.cpu().numpy(), bits, self.n_cols, *geo)
            dev = self.col.device
            hit = (torch.from_numpy(pl.wave_off.view("int32")).to(dev), torch.from_numpy(pl.ent.view("int32")).to(dev), geo)
            self._plans[geo] = hit
        return hit


def spmm_bf16(adj, x, act=ACT_NONE, out=None, out_dtype=torch.bfloat16, path=None, variant=0, geometry=None):
    """act(adj @ x): adj a SparseAdjBf16, x [n_cols, F] bf16 -> [n_rows, F] bf16 (or fp32), fp32 accumulation.
    path: None (by density) | "direct" | "tiled"."""
    if not isinstance(adj, SparseAdjBf16):
        raise TypeError("adj must be a SparseAdjBf16")
    _chk(x, "x", torch.bfloat16, 2)
    if x.shape[0] != adj.n_cols:
        raise ValueError("x has %d rows, the adjacency %d columns" % (x.shape[0], adj.n_cols))
    F = x.shape[1]
    if out is None:
        y = torch.empty(adj.n_rows, F, device=x.device, dtype=out_dtype)
    else:
        y = out
        if y.dtype not in (torch.bfloat16, torch.float32) or tuple(y.shape) != (adj.n_rows, F) or not y.is_contiguous() \
                or not y.is_cuda or y.data_ptr() == x.data_ptr():
            raise ValueError("out must be a contiguous [%d, %d] bf16 / fp32 device tensor distinct from x" % (adj.n_rows, F))
    ybf = 1 if y.dtype == torch.bfloat16 else 0
    if path is None:
        path = "tiled" if (adj.avg_nnz >= SparseAdjBf16.TILED_MIN_AVG_NNZ and F % 256 == 0) else "direct"
    L = _lib.lib()
    if path == "tiled":
        wo, ent, geo = adj.plan(F, geometry)
        _launch("mgnns_spmm_tiled_bf16_fwd", ("mgnns_spmm_tiled_bf16_fwd", adj.n_rows, F), L.mgnns_spmm_tiled_bf16_fwd,
                _p(wo), _p(ent), geo[0], geo[1], geo[2], adj.n_rows, adj.n_cols, _p(x), F, _p(y), ybf, act, _stream())
    elif path == "direct":
        rp, col, val, rmap = (adj.sorted_for(F) if int(variant) == 0 else None) or (adj.row_ptr, adj.col, adj.val, None)
        _launch("mgnns_spmm_csr_bf16_fwd", ("mgnns_spmm_csr_bf16_fwd", adj.n_rows, F), L.mgnns_spmm_csr_bf16_fwd,
                _p(rp), _p(col), _p(val), adj.n_rows, adj.nnz, _p(x), F, _p(y), ybf, act, int(variant), _p(rmap), _stream())
    else:
        raise ValueError("path must be None, 'direct' or 'tiled'")
    return y


# ---- gathers -------------------------------------------------------------------------------
def embedding(idx, table):
    _chk(idx, "idx", torch.int64)
    _chk(table, "table", ndim=2)
    out = torch.empty(*idx.shape, table.shape[1], device=table.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.mgnns_embedding_fwd(_p(idx), idx.numel(), _p(table), table.shape[0], table.shape[1], _p(out),
                                     _stream()), "mgnns_embedding_fwd")
    return out


# ---- text memory bank: embedding + packed BiLSTM -----------------------------------------------------
class LstmCache:
    """Derived forms of one nn.LSTM's weights (per-layer [W_ih ; W_ih_reverse], the packed bf16 layouts), owned by the
    module that owns the weights so their lifetime is the module's (a captured hipGraph bakes these pointers in).
    An entry keeps strong REFERENCES to the tensors it was derived from, so their storage cannot be freed and handed
    to another model's weights while the entry lives: a (data_ptr, version) match therefore means the same weights."""

    def __init__(self):
        self.cat = None        # (sources, versions, value)
        self.prepack = None
        self.table = None      # the layer-0 input projection folded into the embedding table

    @staticmethod
    def _hit(entry, sources):
        return (entry is not None and len(entry[0]) == len(sources)
                and all(a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.device == b.device
                        for a, b in zip(entry[0], sources))
                and entry[1] == tuple(t._version for t in sources))


def _lstm_cat(weights, num_layers, cache):
    """Per layer [W_ih ; W_ih_reverse] and [b_ih ; b_ih_reverse], rebuilt when a weight changes (one input-projection
    GEMM per layer instead of two)."""
    src = [t for tup in weights for t in (tup[0], tup[2])]
    if cache is not None and LstmCache._hit(cache.cat, src):
        return cache.cat[2]
    val = [(torch.cat([weights[2 * l][0], weights[2 * l + 1][0]], 0).contiguous(),
            torch.cat([weights[2 * l][2], weights[2 * l + 1][2]], 0).contiguous()) for l in range(num_layers)]
    if cache is not None:
        if cache.cat is not None:
            retire(cache.cat[2])                # a live capture holds the old concatenations' addresses (projection weights / biases)
        cache.cat = (src, tuple(t._version for t in src), val)
    return val


# bf16 recurrence: the layer-0 input projection folded into the embedding table once per weight version (mgnns_bilstm_bf16_fold_embedding:
# [V, 8 * hidden] fp32, 97 MB for V = 20 154); MGNNS_LSTM_FOLD=0: the projection GEMM on the gathered rows in every forward
LSTM_FOLD_EMBEDDING = os.environ.get("MGNNS_LSTM_FOLD", "1") != "0"


def _lstm_table(emb_table, cat0, hidden, cache):
    """table[v] = bf16(emb[v]) . bf16(W_ih0)^T + b_ih0 for every vocabulary entry; rebuilt when the embedding or layer 0's input
    weights change.  Superseded tables go to the retirement list (a captured graph may still read them)."""
    src = [emb_table, cat0[0], cat0[1]]
    if cache is not None and LstmCache._hit(cache.table, src):
        return cache.table[2]
    L = _lib.lib()
    V, E = emb_table.shape
    table = torch.empty(V, 8 * hidden, device=emb_table.device, dtype=torch.float32)
    assert table.numel() * 4 == L.mgnns_bilstm_bf16_table_bytes(V, hidden)
    ws = torch.empty(L.mgnns_bilstm_bf16_fold_workspace_bytes(V), dtype=torch.uint8, device=emb_table.device)
    _lib.check(L.mgnns_bilstm_bf16_fold_embedding(_p(emb_table), V, E, hidden, _p(cat0[0]), _p(cat0[1]), _p(ws), ws.numel(), _p(table),
                                                  _stream()), "mgnns_bilstm_bf16_fold_embedding")
    if cache is not None:
        if cache.table is not None:
            retire(cache.table[2])
        cache.table = (src, tuple(t._version for t in src), table)
    return table


def bilstm_can_plan(B, T, emb_dim, recurrence="bf16"):
    """Can bilstm(..., plan_mask=...) build the packing plan of the text mask inside its prep launch?"""
    return recurrence == "bf16" and B <= 1024 and T <= PLAN_MAX_L and emb_dim % 4 == 0 and emb_dim <= 320


def bilstm(tok, lens, emb_table, weights, hidden, num_layers, want_bf16=False, recurrence="f32", cache=None, fold=None,
           plan_mask=None):
    """tok [B,T] int64, lens [B] int64 (device), weights = list over (layer, direction) of
    (w_ih, w_hh, b_ih, b_hh) -> [B,T,2*hidden] with zeros behind each sample's length
    (+ the same bank as zero-padded bf16 [B,T,320] when want_bf16).  recurrence="bf16": W_hh . h of every step on the
    bf16 MFMA (bf16 operands, fp32 accumulation and state) instead of the exact fp32 GEMV.
    cache: an LstmCache owned by the module that owns `weights` (None: derived weight forms are rebuilt per call).
    fold (bf16 recurrence; default LSTM_FOLD_EMBEDDING when a cache is given): read the layer-0 input projection out of the
    table folded from the embedding and W_ih once per weight version -- the same rows bit for bit, no GEMM in front of the first
    recurrence.
    plan_mask (bf16 recurrence, bilstm_can_plan): the batch's text mask [B, T] float -- the packing plan of that mask for the packed
    masked attention launches (== sq_mha_plan(plan_mask)) is built by an extra workgroup of the prep launch and returned as a
    third / second value."""
    import ctypes
    _chk(tok, "text", torch.int64, 2)
    _chk(lens, "text_lens", torch.int64, 1)
    _chk(emb_table, "embedding.weight", ndim=2)
    B, T = tok.shape
    if lens.shape[0] != B:
        raise ValueError("text_lens has %d entries for batch %d" % (lens.shape[0], B))
    if len(weights) != 2 * num_layers:
        raise ValueError("need %d (layer, direction) weight tuples" % (2 * num_layers))
    for li, tup in enumerate(weights):
        in_dim = emb_table.shape[1] if li < 2 else 2 * hidden
        shapes = ((4 * hidden, in_dim), (4 * hidden, hidden), (4 * hidden,), (4 * hidden,))
        for t, shp in zip(tup, shapes):
            _chk(t, "lstm weight")
            if tuple(t.shape) != shp:
                raise ValueError("lstm weight shape %s, expected %s" % (tuple(t.shape), shp))
    cat = _lstm_cat(weights, num_layers, cache)
    arr = lambda ptrs: (ctypes.c_void_p * len(ptrs))(*ptrs)
    c_wih = arr([c[0].data_ptr() for c in cat])
    c_bih = arr([c[1].data_ptr() for c in cat])
    c_whh = arr([w[1].data_ptr() for w in weights])
    c_bhh = arr([w[3].data_ptr() for w in weights])
    L = _lib.lib()
    nbytes = L.mgnns_bilstm_workspace_bytes(B, T, hidden, num_layers)
    # per call, on the calling stream: a captured forward keeps it alive in the graph's own pool; nothing is shared
    # between models, streams or graphs
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=tok.device)
    out = torch.empty(B, T, 2 * hidden, device=tok.device, dtype=torch.float32)
    out_bf = torch.empty(B, T, BANK_LD, device=tok.device, dtype=torch.bfloat16) if want_bf16 else None
    plan = None
    if plan_mask is not None:
        _chk(plan_mask, "plan_mask", ndim=2)
        if tuple(plan_mask.shape) != (B, T) or not bilstm_can_plan(B, T, emb_table.shape[1], recurrence):
            raise ValueError("plan_mask must be the [%d, %d] text mask and needs bilstm_can_plan(...)" % (B, T))
        plan = torch.empty(L.mgnns_sq_mha32_plan_ints(B), dtype=torch.int32, device=tok.device)
    if recurrence not in ("f32", "bf16"):
        raise ValueError("recurrence must be 'f32' or 'bf16', got %r" % (recurrence,))
    if recurrence == "f32":
        _launch("mgnns_bilstm_fwd", ("mgnns_bilstm_fwd",), L.mgnns_bilstm_fwd, _p(tok), _p(lens), B, T, _p(emb_table),
                emb_table.shape[0], emb_table.shape[1], hidden, num_layers, c_wih, c_bih, c_whh, c_bhh,
                _p(ws), ws.numel(), _p(out), _p(out_bf), BANK_LD, _stream())
    else:
        # weight layouts of the bf16 kernels: packed once per weight version, off the per-forward path
        src = [t for tup in weights for t in (tup[0], tup[1])]
        if cache is not None and LstmCache._hit(cache.prepack, src):
            pre = cache.prepack[2]
        else:
            pre = torch.empty(L.mgnns_bilstm_bf16_prepack_bytes(hidden, num_layers), dtype=torch.uint8, device=tok.device)
            _lib.check(L.mgnns_bilstm_bf16_prepack(c_wih, c_whh, emb_table.shape[1], hidden, num_layers, _p(pre), _stream()),
                       "mgnns_bilstm_bf16_prepack")
            if cache is not None:
                if cache.prepack is not None:
                    retire(cache.prepack[2])
                cache.prepack = (src, tuple(t._version for t in src), pre)
        if fold is None:
            fold = LSTM_FOLD_EMBEDDING and cache is not None
        if fold and B <= 1024 and T <= 1024 and emb_table.shape[1] % 4 == 0 and emb_table.shape[1] <= 320:
            table = _lstm_table(emb_table, cat[0], hidden, cache)
            _launch("mgnns_bilstm_bf16_table_fwd", ("mgnns_bilstm_bf16_table_fwd",), L.mgnns_bilstm_bf16_table_fwd, _p(tok), _p(lens), B, T,
                    _p(emb_table), emb_table.shape[0], emb_table.shape[1], hidden, num_layers, c_wih, c_bih, c_whh, c_bhh,
                    _p(ws), ws.numel(), _p(out), _p(out_bf), BANK_LD, _p(pre), _p(table), _p(plan_mask), _p(plan), _stream())
        else:
            _launch("mgnns_bilstm_bf16_fwd", ("mgnns_bilstm_bf16_fwd",), L.mgnns_bilstm_bf16_fwd, _p(tok), _p(lens), B, T, _p(emb_table),
                    emb_table.shape[0], emb_table.shape[1], hidden, num_layers, c_wih, c_bih, c_whh, c_bhh,
                    _p(ws), ws.numel(), _p(out), _p(out_bf), BANK_LD, _p(pre), _p(plan_mask), _p(plan), _stream())
    if plan is not None:
        plan._mg_plan_kind = 'packed'
        return (out, out_bf, plan) if want_bf16 else (out, plan)
    return (out, out_bf) if want_bf16 else out


# ---- text GCN --------------------------------------------------------------------------------
def textgcn(tok, node_hidden, edge_w, pmi_dev, ngram, max_length=100):
    """Text_GCN.Model.forward (Text_GCN.py:213-275): tok [B,T] int64 -> [B,D]."""
    _chk(tok, "doc_ids", torch.int64, 2)
    _chk(node_hidden, "node_hidden.weight", ndim=2)
    ew = _chk(edge_w.reshape(-1), "seq_edge_w.weight")
    rp, col, eid = pmi_dev
    _chk(rp, "pmi row_ptr", torch.int32, 1)
    _chk(col, "pmi col", torch.int32, 1)
    if eid is not None:                    # None: ids are positional (eid[k] == k + 1), the kernel derives them
        _chk(eid, "pmi eid", torch.int32, 1)
        if eid.shape != col.shape:
            raise ValueError("pmi eid and col must have the same length")
    B, T = tok.shape
    V, D = node_hidden.shape
    if rp.shape[0] != V + 1:
        raise ValueError("PMI map has %d rows, vocabulary has %d" % (rp.shape[0] - 1, V))
    out = torch.empty(B, D, device=tok.device, dtype=torch.float32)
    L = _lib.lib()
    _launch("mgnns_textgcn_fwd", ("mgnns_textgcn_fwd",), L.mgnns_textgcn_fwd, _p(tok), B, T, _p(node_hidden), V, D,
            _p(ew), ew.shape[0], _p(rp), _p(col), _p(eid), int(ngram), int(max_length), _p(out), _stream())
    return out


# ---- image bank + pool ---------------------------------------------------------------------------
IMGBANK_LDW = 304


def transpose_pad(w, ld):
    """[rows, cols] -> [cols, ld] transposed, zero padded."""
    _chk(w, "w", ndim=2)
    out = torch.empty(w.shape[1], ld, device=w.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.mgnns_transpose_pad(_p(w), w.shape[0], w.shape[1], _p(out), ld, _stream()), "mgnns_transpose_pad")
    return out


def imgbank_pool(feat, wt, bias, n_out, want_pool=True):
    """feat [B,K,P]; wt = transpose_pad(liner_img.weight, 304) [K,304] -> bank [B,P,N], pooled [B,K]."""
    _chk(feat, "feature map", ndim=3)
    _chk(wt, "wt", ndim=2)
    B, K, P = feat.shape
    if wt.shape[0] != K:
        raise ValueError("wt rows %d != K %d" % (wt.shape[0], K))
    if bias is not None:
        _chk(bias, "bias", ndim=1)
    bank = torch.empty(B, P, n_out, device=feat.device, dtype=torch.float32)
    pooled = torch.empty(B, K, device=feat.device, dtype=torch.float32) if want_pool else None
    L = _lib.lib()
    _launch("mgnns_imgbank_pool_fwd", ("mgnns_imgbank_pool_fwd",), L.mgnns_imgbank_pool_fwd, _p(feat), B, K, P,
            _p(wt), wt.shape[1], _p(bias), n_out, _p(bank), _p(pooled), _stream())
    return bank, pooled


def pack_imgbank_weights_bf16(w):
    """liner_img_*.weight [N, K] fp32 -> MFMA-fragment-major bf16 buffer for imgbank_pool_bf16."""
    _chk(w, "weight", ndim=2)
    L = _lib.lib()
    buf = torch.empty(L.mgnns_imgbank_packed_weight_bytes(w.shape[1]), dtype=torch.uint8, device=w.device)
    _lib.check(L.mgnns_imgbank_pack_weights_bf16(_p(w), w.shape[0], w.shape[1], _p(buf), _stream()),
               "mgnns_imgbank_pack_weights_bf16")
    return buf


def imgbank_pool_bf16(feat, wp, bias, n_out, combine=True):
    """feat [B,K,P] fp32 -> (bank bf16 [B,P,320], pooled fp32 [B,K]); combine=False: pooled stays as the kernel's two
    per-half maxima [B,2,K] (label_tail takes the max itself: one launch less on the channel's critical chain)."""
    _chk(feat, "feature map", ndim=3)
    _chk(wp, "packed weight", torch.uint8, 1)
    B, K, P = feat.shape
    if bias is not None:
        _chk(bias, "bias", ndim=1)
    bank = torch.empty(B, P, BANK_LD, device=feat.device, dtype=torch.bfloat16)
    pooled = torch.empty(B, K, device=feat.device, dtype=torch.float32)
    work = torch.empty(B, 2, K, device=feat.device, dtype=torch.float32)
    L = _lib.lib()
    _launch("mgnns_imgbank_pool_bf16_fwd", ("mgnns_imgbank_pool_bf16_fwd",), L.mgnns_imgbank_pool_bf16_fwd, _p(feat),
            B, K, P, _p(wp), _p(bias), n_out, _p(bank), BANK_LD, _p(pooled) if combine else None, _p(work), _stream())
    return bank, (pooled if combine else work)


def head_diff(o, n_head):
    """o [B, n_head*d_v] (per-head attention outputs) -> [B]: mean over head pairs i != j of cos^2(o_i, o_j)
    (diff_outputs, submodules.py:38-52)."""
    _chk(o, "attention output", ndim=2)
    B = o.shape[0]
    if o.shape[1] % n_head:
        raise ValueError("attention output width %d is not a multiple of n_head=%d" % (o.shape[1], n_head))
    out = torch.empty(B, device=o.device, dtype=torch.float32)
    L = _lib.lib()
    _launch("mgnns_head_diff_fwd", ("mgnns_head_diff_fwd",), L.mgnns_head_diff_fwd, _p(o), B, n_head, o.shape[1] // n_head, _p(out),
            _stream())
    return out


def textgcn_set_form(form):
    """0 by batch (default) | 1 one 1024-thread launch | 2 two launches (short + long documents) | 3 the lean kernel (tests)."""
    _lib.check(_lib.lib().mgnns_textgcn_set_form(int(form)), "mgnns_textgcn_set_form")


def imgbank_set_form(form):
    """Which bf16 image-bank kernel runs: 0 = chosen by the batch (default), 1 = the stream form (one workgroup per sample),
    2 = the pair form (two workgroups per sample) where its shape limits allow.  Process-wide; tests and measurements."""
    _lib.check(_lib.lib().mgnns_imgbank_set_form(int(form)), "mgnns_imgbank_set_form")


# ---- label attention core ---------------------------------------------------------------------------
def imgbank_pool_split(feat, w_pair, bias, n_out, want_pool=True, want_f32=True, want_split=False):
    """Split-bf16 (fp32-class) image bank + max-pool: feat [B,K,P] fp32, w_pair = pack_weight_bf16_split(liner_img.weight
    [n_out,K]) -> (bank [B,P,n_out] fp32 or None, pooled halves [B,2,K] fp32 or None[, split images bf16 [2,B,P,320] when
    want_split: hi = bf16(bank), lo = bf16(bank - hi), zero padded -- the split-bf16 attention core's operand])."""
    _chk(feat, "feature map", ndim=3)
    B, K, P = feat.shape
    hi, lo = w_pair
    _chk(hi, "packed hi", torch.uint8, 1)
    _chk(lo, "packed lo", torch.uint8, 1)
    if bias is not None:
        _chk(bias, "bias", ndim=1)
    if not (want_f32 or want_split):
        raise ValueError("imgbank_pool_split: nothing to compute (want_f32 or want_split)")
    bank = torch.empty(B, P, n_out, device=feat.device, dtype=torch.float32) if want_f32 else None
    split = torch.empty(2, B, P, BANK_LD, device=feat.device, dtype=torch.bfloat16) if want_split else None
    pooled = torch.empty(B, 2, K, device=feat.device, dtype=torch.float32) if want_pool else None
    L = _lib.lib()
    _launch("mgnns_imgbank_pool_split_fwd", ("mgnns_imgbank_pool_split_fwd", P), L.mgnns_imgbank_pool_split_fwd, _p(feat), B, K, P,
            _p(hi), _p(lo), _p(bias), n_out, _p(bank), _p(pooled), _p(split[0]) if want_split else None,
            _p(split[1]) if want_split else None, _stream())
    return (bank, pooled, split) if want_split else (bank, pooled)


def label_attn_core(Q, K, V, n_heads, mask=None):
    """x[b,l,:] of Attention.forward between its projections (MODEL:101-131).  mask: anything that broadcasts against the
    energy [B, NLQ, heads, dh] (MODEL:118-119: positions where mask == 0 get -1e10 in front of the softmax over dh)."""
    _chk(Q, "Q", ndim=2)
    _chk(K, "K", ndim=2)
    _chk(V, "V", ndim=2)
    NLQ, hid = Q.shape
    B = K.shape[0]
    if K.shape[1] != hid or V.shape != K.shape or hid % n_heads:
        raise ValueError("label attention shapes Q%s K%s V%s" % (tuple(Q.shape), tuple(K.shape), tuple(V.shape)))
    x = torch.empty(B, NLQ, hid, device=K.device, dtype=torch.float32)
    L = _lib.lib()
    if mask is not None:
        if mask.device != K.device:
            raise ValueError("mask is on %s, K on %s" % (mask.device, K.device))
        # the byte image of the broadcast mask (layout only: the comparison with 0 and the fill run in the kernel)
        m8 = torch.broadcast_to(mask != 0, (B, NLQ, n_heads, hid // n_heads)).to(torch.uint8).contiguous()
        _lib.check(L.mgnns_label_attn_core_masked_fwd(_p(Q), _p(K), _p(V), _p(m8), B, NLQ, n_heads, hid // n_heads, _p(x), _stream()),
                   "mgnns_label_attn_core_masked_fwd")
        return x
    _lib.check(L.mgnns_label_attn_core_fwd(_p(Q), _p(K), _p(V), B, NLQ, n_heads, hid // n_heads, _p(x), _stream()),
               "mgnns_label_attn_core_fwd")
    return x


def label_tail(x, Q, n_heads, packed, pooled=None, g_wp=None, next_q=None):
    """Fused label-attention tail (mgnns_label_tail_fwd).  Either x [B,C] (the read-out) or pooled [B,parts,K] +
    g_wp = pack_weight_f32(G [C,K]) (read-out computed in the kernel); Q [NLQ,hid] = w_q(label query); packed =
    dict(wk, bk, wv, bv, wc, bc, n5, xl, bxl, n_out, C) with wk/wv/wc/xl = pack_weight_f32 images; next_q = (wq_wp, bq,
    HK) adds qh = w_qs(out) + b.  -> out [B, n_out]  (or (out, qh) with next_q)."""
    _chk(Q, "Q", ndim=2)
    NLQ, hid = Q.shape
    if hid % n_heads:
        raise ValueError("hidden width %d is not a multiple of %d heads" % (hid, n_heads))
    if (x is None) == (g_wp is None):
        raise ValueError("pass either x or (pooled, g_wp)")
    parts = kp = 0
    if x is not None:
        _chk(x, "x", ndim=2)
        B, C = x.shape
    else:
        _chk(pooled, "pooled", ndim=3)
        _chk(g_wp, "packed G", ndim=1)
        B, parts, kp = pooled.shape
        C = packed["C"]
    if C != packed["C"]:
        raise ValueError("read-out width %d does not match the packed w_k / w_v (%d)" % (C, packed["C"]))
    for k in ("wk", "wv", "wc", "xl", "bk", "bv", "bc", "bxl"):
        _chk(packed[k], k, ndim=1)
    out = torch.empty(B, packed["n_out"], device=Q.device, dtype=torch.float32)
    wq = bq = qh = None
    hkn = 0
    if next_q is not None:
        wq, bq, hkn = next_q
        qh = torch.empty(B, hkn, device=Q.device, dtype=torch.float32)
    L = _lib.lib()
    _launch("mgnns_label_tail_fwd", ("mgnns_label_tail_fwd", C), L.mgnns_label_tail_fwd, _p(x), B, C, _p(pooled), parts, kp,
            _p(g_wp), _p(Q), NLQ, n_heads, hid // n_heads, _p(packed["wk"]), _p(packed["bk"]), _p(packed["wv"]),
            _p(packed["bv"]), _p(packed["wc"]), _p(packed["bc"]), packed["n5"], _p(packed["xl"]), _p(packed["bxl"]),
            packed["n_out"], _p(out), _p(wq), _p(bq), hkn, _p(qh), _stream())
    return out if next_q is None else (out, qh)


LABEL_TAIL_CLUSTER = os.environ.get("MGNNS_LABEL_TAIL_CLUSTER", "1") != "0"


def label_tail_bf16(pooled, g_pair, Q, n_heads, packed, next_q=None, terms=3, cluster=None):
    """bf16-mode fused channel tail (mgnns_label_tail_bf16_fwd): pooled [B,parts,K] fp32, g_pair = pack_weight_bf16_split(G
    [C,K]); packed = dict(wk, wv, wc, xl = (hi, lo) pairs of pack_weight_bf16_split, bk, bv, bc, bxl, n5, n_out, C);
    next_q = ((hi, lo), bq, HK).  terms = 1 plain bf16 | 3 split-bf16.  cluster: four workgroups per 16-sample tile
    (terms = 3; default on, MGNNS_LABEL_TAIL_CLUSTER=0 turns it off); the exchange scratch lives in `packed`, one per
    channel, so the two channels' launches never share one.  -> out [B,n_out] (or (out, qh))."""
    import ctypes
    _chk(pooled, "pooled", ndim=3)
    _chk(Q, "Q", ndim=2)
    B, parts, kp = pooled.shape
    NLQ, hid = Q.shape
    if hid % n_heads:
        raise ValueError("hidden width %d is not a multiple of %d heads" % (hid, n_heads))
    pairs = [g_pair, packed["wk"], packed["wv"], packed["wc"], packed["xl"]]
    for k in ("bk", "bv", "bc", "bxl"):
        _chk(packed[k], k, ndim=1)
    out = torch.empty(B, packed["n_out"], device=Q.device, dtype=torch.float32)
    bq = qh = None
    hkn = 0
    if next_q is not None:
        wqp, bq, hkn = next_q
        pairs.append(wqp)
        qh = torch.empty(B, hkn, device=Q.device, dtype=torch.float32)
    ptrs = []
    for h, l in pairs:
        _chk(h, "packed hi", torch.uint8, 1)
        _chk(l, "packed lo", torch.uint8, 1)
        ptrs += [h.data_ptr(), l.data_ptr()]
    ptrs += [None] * (12 - len(ptrs))
    arr = (ctypes.c_void_p * 12)(*ptrs)
    if cluster is None:
        cluster = LABEL_TAIL_CLUSTER
    scratch = counters = None
    if cluster and int(terms) == 3 and B > 0:
        tiles = (B + 15) // 16
        slot = packed.setdefault("_cluster_ws", {})
        key = _scratch_key()                                     # per (capture epoch, launch stream), like the label GCN's
        ws = slot.get(key)
        if ws is None or ws[0] < tiles or ws[1].device != Q.device:
            if ws is not None:
                packed.setdefault("_retired", []).append(ws)     # a captured hipGraph may still hold its address: never freed
            ws = (tiles, torch.empty(tiles * 4 * 6144, device=Q.device, dtype=torch.float32),
                  zeros_i32(2 * tiles, Q.device))
            _scratch_slot_put(slot, key, ws)
        scratch, counters = ws[1], ws[2]
    L = _lib.lib()
    _launch("mgnns_label_tail_bf16_fwd", ("mgnns_label_tail_bf16_fwd", packed["C"]), L.mgnns_label_tail_bf16_fwd, _p(pooled), B,
            parts, kp, packed["C"], int(terms), arr, _p(Q), NLQ, n_heads, hid // n_heads, _p(packed["bk"]), _p(packed["bv"]),
            _p(packed["bc"]), packed["n5"], _p(packed["bxl"]), packed["n_out"], _p(out), _p(bq), hkn, _p(qh), _p(scratch),
            _p(counters), _stream())
    return out if next_q is None else (out, qh)


# ---- single-query MHA core -------------------------------------------------------------------------------
def sq_mha_core(qh, bank, mask, n_head, d_kv, wk, bk, wv, bv, want_attn=True):
    _chk(qh, "qh", ndim=2)
    _chk(bank, "memory bank", ndim=3)
    B, L_, D = bank.shape
    if qh.shape != (B, n_head * d_kv):
        raise ValueError("qh shape %s, expected %s" % (tuple(qh.shape), (B, n_head * d_kv)))
    if mask is not None:
        _chk(mask, "mask", ndim=2)
        if mask.shape != (B, L_):
            raise ValueError("mask shape %s, expected %s" % (tuple(mask.shape), (B, L_)))
    for n, t in (("w_ks.weight", wk), ("w_vs.weight", wv)):
        _chk(t, n, ndim=2)
        if t.shape != (n_head * d_kv, D):
            raise ValueError("%s shape %s" % (n, tuple(t.shape)))
    o = torch.empty(B, n_head * d_kv, device=bank.device, dtype=torch.float32)
    attn = torch.empty(n_head * B, 1, L_, device=bank.device, dtype=torch.float32) if want_attn else None
    L = _lib.lib()
    _launch("mgnns_sq_mha_core_fwd", ("mgnns_sq_mha_core_fwd", L_, mask is not None), L.mgnns_sq_mha_core_fwd,
            _p(qh), _p(bank), _p(mask), B, L_, D, n_head, d_kv, _p(wk), _p(bk), _p(wv), _p(bv), _p(o), _p(attn),
            _stream())
    return o, attn


# The bf16 attention core has two builds: 32 = v_mfma_f32_32x32x16_bf16 (csrc/sq_mha32_bf16.hip; packed masked banks), 16 = the
# 16x16x32 form of rounds 1-3 (csrc/sq_mha_bf16.hip; also what the fused layer kernel runs).  The packed weights differ: a pack
# carries its form as an attribute and sq_mha_core_bf16 dispatches on it.
MHA_CORE = int(os.environ.get("MGNNS_MHA_CORE", "32"))
MHA_CORE_PLAIN = int(os.environ.get("MGNNS_MHA_CORE_PLAIN", "16"))      # one workgroup per sample (no plan): the faster build
MHA_PACKED = os.environ.get("MGNNS_MHA_PACKED", "1") == "1"      # masked banks: pack the live rows of short samples (a plan)


def pack_kv_weights_bf16(wk, wv, n_head, d_kv, form=None):
    """w_ks / w_vs [H*dk, D] fp32 -> MFMA-fragment-major bf16 buffer for sq_mha_core_bf16."""
    _chk(wk, "w_ks.weight", ndim=2)
    _chk(wv, "w_vs.weight", ndim=2)
    form = MHA_CORE if form is None else int(form)
    L = _lib.lib()
    if form == 32:
        buf = torch.empty(L.mgnns_sq_mha32_packed_weight_bytes(n_head), dtype=torch.uint8, device=wk.device)
        _lib.check(L.mgnns_sq_mha32_pack_weights_bf16(_p(wk), _p(wv), n_head, d_kv, wk.shape[1], _p(buf), _stream()),
                   "mgnns_sq_mha32_pack_weights_bf16")
    else:
        buf = torch.empty(L.mgnns_sq_mha_packed_weight_bytes(n_head), dtype=torch.uint8, device=wk.device)
        _lib.check(L.mgnns_sq_mha_pack_weights_bf16(_p(wk), _p(wv), n_head, d_kv, wk.shape[1], _p(buf), _stream()),
                   "mgnns_sq_mha_pack_weights_bf16")
    buf._mg_form = form
    return buf


def sq_mha_plan(mask):
    """Packing plan of a [B, L] mask (L <= 128) for sq_mha_core_bf16(plan=...): which samples share a workgroup.  One launch;
    build it once per batch, every attention launch on that mask takes it."""
    _chk(mask, "mask", ndim=2)
    B, L_ = mask.shape
    if B > PLAN_MAX_B:
        raise ValueError("sq_mha_plan: batch %d beyond the plan kernel's limit %d (fusion.make_mask_plan falls back to the "
                         "one-workgroup-per-sample core)" % (B, PLAN_MAX_B))
    L = _lib.lib()
    plan = torch.empty(L.mgnns_sq_mha32_plan_ints(B), dtype=torch.int32, device=mask.device)
    _lib.check(L.mgnns_sq_mha32_plan(_p(mask), B, L_, _p(plan), _stream()), "mgnns_sq_mha32_plan")
    plan._mg_plan_kind = 'packed'
    return plan


PLAN_MAX_L = 128
PLAN_MAX_B = 4096          # mgnns_sq_mha32_plan: one workgroup scans the batch ((8 B + 4) * 4 bytes of LDS)


def cast_pad_bf16(x, ld=BANK_LD):
    """[..., D] fp32 -> [..., ld] bf16 (round to nearest even), zero padded."""
    D = x.shape[-1]
    x2 = _chk(x.reshape(-1, D), "x")
    y = torch.empty(x2.shape[0], ld, dtype=torch.bfloat16, device=x.device)
    L = _lib.lib()
    _lib.check(L.mgnns_cast_pad_bf16(_p(x2), x2.shape[0], D, ld, _p(y), _stream()), "mgnns_cast_pad_bf16")
    return y.view(*x.shape[:-1], ld)


def sq_mha_core_bf16(qh, bank_bf16, mask, n_head, d_kv, wp, bk, bv, want_attn=True, plan=None):
    """plan: sq_mha_plan(mask) (32x32x16 form, L <= 128) -- the live rows of short samples packed into shared workgroups."""
    _chk(qh, "qh", ndim=2)
    _chk(bank_bf16, "memory bank (bf16)", torch.bfloat16, 3)
    _chk(wp, "packed K/V weights", torch.uint8, 1)
    B, L_, ld = bank_bf16.shape
    if qh.shape != (B, n_head * d_kv):
        raise ValueError("qh shape %s, expected %s" % (tuple(qh.shape), (B, n_head * d_kv)))
    if mask is not None:
        _chk(mask, "mask", ndim=2)
        if mask.shape != (B, L_):
            raise ValueError("mask shape %s, expected %s" % (tuple(mask.shape), (B, L_)))
    o = torch.empty(B, n_head * d_kv, device=qh.device, dtype=torch.float32)
    attn = torch.empty(n_head * B, 1, L_, device=qh.device, dtype=torch.float32) if want_attn else None
    L = _lib.lib()
    form = getattr(wp, "_mg_form", 16)
    if form == 32:
        if plan is not None:
            _chk(plan, "plan", torch.int32, 1)
            # (4 + 6 B ints: the size names the batch the plan was built for -- a plan of another batch is refused here)
            if mask is None or L_ > PLAN_MAX_L or plan.numel() != L.mgnns_sq_mha32_plan_ints(B):
                raise ValueError("a packing plan needs a mask, L <= %d and exactly %d ints (a plan built for this batch size)"
                                 % (PLAN_MAX_L, L.mgnns_sq_mha32_plan_ints(B)))
            if getattr(plan, "_mg_plan_kind", 'packed') != 'packed':       # (the grouped split-bf16 core's plan has the same size)
                raise ValueError("sq_mha_core_bf16 takes the plan sq_mha_plan(mask) / bilstm(plan_mask=...) returned (got kind %r)"
                                 % plan._mg_plan_kind)
        _launch("mgnns_sq_mha_core_bf16_fwd", ("mgnns_sq_mha_core_bf16_fwd", L_, mask is not None),
                L.mgnns_sq_mha32_core_bf16_fwd, _p(qh), _p(bank_bf16), _p(mask), B, L_, ld, n_head, d_kv, _p(wp), _p(bk),
                _p(bv), _p(o), _p(attn), _p(plan), _stream())
        return o, attn
    if plan is not None:
        raise ValueError("a packing plan needs the 32x32x16 form of the packed weights")
    _launch("mgnns_sq_mha_core_bf16_fwd", ("mgnns_sq_mha_core_bf16_fwd", L_, mask is not None),
            L.mgnns_sq_mha_core_bf16_fwd, _p(qh), _p(bank_bf16), _p(mask), B, L_, ld, n_head, d_kv, _p(wp), _p(bk),
            _p(bv), _p(o), _p(attn), _stream())
    return o, attn


def pack_kv_weights_split(wk, wv, n_head, d_kv):
    """w_ks / w_vs [H*dk, D] fp32 -> the split-bf16 (hi image, lo image) fragment-major buffer of sq_mha_core_split."""
    _chk(wk, "w_ks.weight", ndim=2)
    _chk(wv, "w_vs.weight", ndim=2)
    L = _lib.lib()
    buf = torch.empty(L.mgnns_sq_mha_split_packed_weight_bytes(n_head), dtype=torch.uint8, device=wk.device)
    _lib.check(L.mgnns_sq_mha_pack_weights_split(_p(wk), _p(wv), n_head, d_kv, wk.shape[1], _p(buf), _stream()),
               "mgnns_sq_mha_pack_weights_split")
    buf._mg_form = "split"
    return buf


def split_pad_bf16(x, ld=BANK_LD):
    """[..., D] fp32 -> bf16 [2, ..., ld]: [0] = hi = bf16(x), [1] = lo = bf16(x - hi), zero padded (x = hi + lo to ~2^-17)."""
    D = x.shape[-1]
    x2 = _chk(x.reshape(-1, D), "x")
    y = torch.empty(2, x2.shape[0], ld, dtype=torch.bfloat16, device=x.device)
    L = _lib.lib()
    _lib.check(L.mgnns_split_pad_bf16(_p(x2), x2.shape[0], D, ld, _p(y[0]), _p(y[1]), _stream()), "mgnns_split_pad_bf16")
    return y.view(2, *x.shape[:-1], ld)


SPLIT_PLAN_MAX_L = 112


def sq_mha_split_plan(mask):
    """Group plan of a [B, L] mask (L <= 112) for sq_mha_core_split(plan=...): which samples share a workgroup (whole 16-row tiles,
    at most 7 tiles and 7 samples per group).  One launch; once per batch."""
    _chk(mask, "mask", ndim=2)
    B, L_ = mask.shape
    if B > PLAN_MAX_B or L_ > SPLIT_PLAN_MAX_L:
        raise ValueError("sq_mha_split_plan: B=%d (<= %d), L=%d (<= %d)" % (B, PLAN_MAX_B, L_, SPLIT_PLAN_MAX_L))
    L = _lib.lib()
    plan = torch.empty(L.mgnns_sq_mha32_plan_ints(B), dtype=torch.int32, device=mask.device)
    _lib.check(L.mgnns_sq_mha_split_plan(_p(mask), B, L_, _p(plan), _stream()), "mgnns_sq_mha_split_plan")
    plan._mg_plan_kind = 'grouped'
    return plan


def sq_mha_core_split(qh, bank_split, mask, n_head, d_kv, wp, bk, bv, want_attn=True, plan=None):
    """The faithful attention core on split-bf16 operands (csrc/sq_mha_split_bf16.hip): bank_split = split_pad_bf16(bank)
    [2, B, L, 320], wp = pack_kv_weights_split(...).  plan: sq_mha_split_plan(mask) (masked banks, L <= 112, no attn output): the
    samples of a group share a workgroup.  -> (o [B, H*dk], attn [H*B, 1, L] or None)"""
    _chk(qh, "qh", ndim=2)
    _chk(bank_split, "memory bank (split bf16)", torch.bfloat16, 4)
    _chk(wp, "packed K/V weights", torch.uint8, 1)
    if getattr(wp, "_mg_form", None) != "split":
        raise ValueError("sq_mha_core_split takes pack_kv_weights_split(...) weights")
    two, B, L_, ld = bank_split.shape
    if two != 2:
        raise ValueError("bank_split must be [2, B, L, %d] (hi, lo)" % BANK_LD)
    if qh.shape != (B, n_head * d_kv):
        raise ValueError("qh shape %s, expected %s" % (tuple(qh.shape), (B, n_head * d_kv)))
    if mask is not None:
        _chk(mask, "mask", ndim=2)
        if mask.shape != (B, L_):
            raise ValueError("mask shape %s, expected %s" % (tuple(mask.shape), (B, L_)))
    o = torch.empty(B, n_head * d_kv, device=qh.device, dtype=torch.float32)
    attn = torch.empty(n_head * B, 1, L_, device=qh.device, dtype=torch.float32) if want_attn else None
    L = _lib.lib()
    if plan is not None:
        _chk(plan, "plan", torch.int32, 1)
        if mask is None or L_ > SPLIT_PLAN_MAX_L or want_attn or plan.numel() != L.mgnns_sq_mha32_plan_ints(B):
            raise ValueError("a group plan needs a mask, L <= %d, want_attn=False and exactly %d ints (a plan built for this batch)"
                             % (SPLIT_PLAN_MAX_L, L.mgnns_sq_mha32_plan_ints(B)))
        # the packed bf16 kernel's plan has the SAME size (up to 16 samples / 128 rows per group: past this kernel's LDS maps); only
        # the tensor sq_mha_split_plan returned is taken (a view / clone loses the tag -- the kernel checks the plan's own header word
        # too and raises the library's status word instead of running)
        if getattr(plan, "_mg_plan_kind", None) != 'grouped':
            raise ValueError("sq_mha_core_split takes the plan sq_mha_split_plan(mask) returned (got kind %r)"
                             % getattr(plan, "_mg_plan_kind", None))
    _launch("mgnns_sq_mha_core_split_fwd", ("mgnns_sq_mha_core_split_fwd", L_, mask is not None), L.mgnns_sq_mha_core_split_fwd,
            _p(qh), _p(bank_split[0]), _p(bank_split[1]), _p(mask), B, L_, ld, n_head, d_kv, _p(wp), _p(bk), _p(bv), _p(o),
            _p(attn), _p(plan), _stream())
    return o, attn


def sq_mha_layer_bf16(qh, bank_bf16, mask, n_head, d_kv, wp, bk, bv, q, packed, eps, counters, next_packed=None):
    """One fusion layer in one launch (mgnns_sq_mha_layer_bf16_fwd): attention core + fused tail (plain bf16 operands).
    q: the layer input [B,300]; packed / next_packed as for mha_tail_bf16; counters: int32 zeros [ceil(B/16)] owned by the
    layer (left zero by the kernel).  -> (out [B,300], qh_next or None)"""
    import ctypes
    _chk(qh, "qh", ndim=2)
    _chk(bank_bf16, "memory bank (bf16)", torch.bfloat16, 3)
    _chk(wp, "packed K/V weights", torch.uint8, 1)
    if getattr(wp, "_mg_form", 16) != 16:
        raise ValueError("the fused layer kernel takes the 16x16x32 form of the packed K/V weights (pack_kv_weights_bf16(form=16))")
    _chk(q, "q", ndim=2)
    _chk(counters, "tile counters", torch.int32, 1)
    B, L_, ld = bank_bf16.shape
    if qh.shape != (B, n_head * d_kv) or q.shape != (B, 300):
        raise ValueError("qh %s / q %s do not match batch %d" % (tuple(qh.shape), tuple(q.shape), B))
    if counters.shape[0] < (B + 15) // 16:
        raise ValueError("need %d tile counters" % ((B + 15) // 16))
    if mask is not None:
        _chk(mask, "mask", ndim=2)
        if mask.shape != (B, L_):
            raise ValueError("mask shape %s, expected %s" % (tuple(mask.shape), (B, L_)))
    o = torch.empty(B, n_head * d_kv, device=qh.device, dtype=torch.float32)
    out = torch.empty(B, 300, device=qh.device, dtype=torch.float32)
    ptrs = [packed["fc"][0].data_ptr(), packed["fc"][1].data_ptr(), packed["w1"][0].data_ptr(), packed["w1"][1].data_ptr(),
            packed["w2"][0].data_ptr(), packed["w2"][1].data_ptr(), None, None]
    bq = qhn = None
    hkn = 0
    if next_packed is not None:
        (wh, wl), bq, hkn = next_packed
        ptrs[6], ptrs[7] = wh.data_ptr(), wl.data_ptr()
        qhn = torch.empty(B, hkn, device=qh.device, dtype=torch.float32)
    arr = (ctypes.c_void_p * 8)(*ptrs)
    L = _lib.lib()
    _launch("mgnns_sq_mha_layer_bf16_fwd", ("mgnns_sq_mha_layer_bf16_fwd", L_, mask is not None), L.mgnns_sq_mha_layer_bf16_fwd,
            _p(qh), _p(bank_bf16), _p(mask), B, L_, ld, n_head, d_kv, _p(wp), _p(bk), _p(bv), _p(o), _p(q), 300, arr,
            _p(packed["fc_b"]), _p(packed["g1"]), _p(packed["be1"]), _p(packed["b1"]), _p(packed["b2"]), _p(packed["g2"]),
            _p(packed["be2"]), float(eps), _p(out), _p(bq), hkn, _p(qhn), _p(counters), _stream())
    return out, qhn


def sq_mha_folded(qh, bank, mask, n_head, d_kv, wk, wv, bv, want_attn=True):
    """Folded single-query attention (mgnns_sq_mha_folded_fwd): same result as sq_mha_core, K and V never formed.
    bank: fp32 [B, L, D] or bf16 [B, L, ld] (zero padded).  b_k is not needed (it drops out of the softmax)."""
    _chk(qh, "qh", ndim=2)
    is_bf16 = bank.dtype == torch.bfloat16
    _chk(bank, "memory bank", torch.bfloat16 if is_bf16 else torch.float32, 3)
    B, L_, ld = bank.shape
    for n, t in (("w_ks.weight", wk), ("w_vs.weight", wv)):
        _chk(t, n, ndim=2)
    D = wk.shape[1]
    if qh.shape != (B, n_head * d_kv):
        raise ValueError("qh shape %s, expected %s" % (tuple(qh.shape), (B, n_head * d_kv)))
    if wk.shape != (n_head * d_kv, D) or wv.shape != (n_head * d_kv, D):
        raise ValueError("w_ks %s / w_vs %s, expected %s" % (tuple(wk.shape), tuple(wv.shape), (n_head * d_kv, D)))
    if (ld != D) if not is_bf16 else (ld < D):
        raise ValueError("memory bank last dim %d does not match d_model %d" % (ld, D))
    if mask is not None:
        _chk(mask, "mask", ndim=2)
        if mask.shape != (B, L_):
            raise ValueError("mask shape %s, expected %s" % (tuple(mask.shape), (B, L_)))
    o = torch.empty(B, n_head * d_kv, device=qh.device, dtype=torch.float32)
    attn = torch.empty(n_head * B, 1, L_, device=qh.device, dtype=torch.float32) if want_attn else None
    L = _lib.lib()
    nbytes = L.mgnns_sq_mha_folded_workspace_bytes(B, D, n_head)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=qh.device)     # per call: stacks run on four streams
    _launch("mgnns_sq_mha_folded_fwd", ("mgnns_sq_mha_folded_fwd", L_, is_bf16), L.mgnns_sq_mha_folded_fwd,
            _p(qh), _p(bank), int(is_bf16), ld, _p(mask), B, L_, D, n_head, d_kv, _p(wk), _p(wv), _p(bv), _p(ws), nbytes,
            _p(o), _p(attn), _stream())
    return o, attn


def sq_mha_folded_bf16(u, bank_bf16, mask, n_head, d_kv, want_attn=True):
    """Folded single-query attention on the bf16 matrix pipe (mgnns_sq_mha_folded_bf16_fwd).  u: fp32 [B, H*D] composed query
    rows (fusion.composed_query_map); bank: bf16 [B, L, 320] zero padded -> (c bf16 [B, H*D rounded up to 32] = per-head
    probability-weighted bank rows at h*D, zeros behind H*D: what mha_tail_c16 takes; attn [H*B, 1, L] or None)."""
    _chk(u, "u", ndim=2)
    _chk(bank_bf16, "memory bank", torch.bfloat16, 3)
    B, L_, ld = bank_bf16.shape
    if ld != 320:
        raise ValueError("bf16 memory bank last dim %d, expected 320 (cast_pad_bf16)" % ld)
    if u.shape[0] != B or u.shape[1] % n_head:
        raise ValueError("u shape %s does not match batch %d / %d heads" % (tuple(u.shape), B, n_head))
    D = u.shape[1] // n_head
    if mask is not None:
        _chk(mask, "mask", ndim=2)
        if mask.shape != (B, L_):
            raise ValueError("mask shape %s, expected %s" % (tuple(mask.shape), (B, L_)))
    ldc = (n_head * D + 31) // 32 * 32
    c = torch.empty(B, ldc, device=u.device, dtype=torch.bfloat16)
    attn = torch.empty(n_head * B, 1, L_, device=u.device, dtype=torch.float32) if want_attn else None
    L = _lib.lib()
    _launch("mgnns_sq_mha_folded_bf16_fwd", ("mgnns_sq_mha_folded_bf16_fwd", L_), L.mgnns_sq_mha_folded_bf16_fwd,
            _p(u), _p(bank_bf16), _p(mask), B, L_, D, n_head, float(1.0 / (d_kv ** 0.5)), _p(c), ldc, _p(attn), _stream())
    return c, attn


def pack_weight_f32(w):
    """[N, K] fp32 (nn.Linear layout) -> MFMA-fragment-major fp32 buffer for mha_tail."""
    _chk(w, "weight", ndim=2)
    L = _lib.lib()
    buf = torch.empty(L.mgnns_packed_f32_weight_bytes(w.shape[0], w.shape[1]) // 4, dtype=torch.float32, device=w.device)
    _lib.check(L.mgnns_pack_weight_f32(_p(w), w.shape[0], w.shape[1], _p(buf), _stream()), "mgnns_pack_weight_f32")
    return buf


def mha_tail(o, q, packed, eps, next_packed=None):
    """Fused fc + residual + LN + FFN + residual + LN (+ next layer's w_qs).  `packed` = dict with fc_wp, fc_b,
    g1, be1, w1_wp, b1, w2_wp, b2, g2, be2; next_packed = (wq_wp, bq, HK_next) or None.
    Returns (out [B,300], qh_next [B,HK_next] or None)."""
    _chk(o, "o", ndim=2)
    _chk(q, "q", ndim=2)
    B, HK = o.shape
    if q.shape != (B, 300):
        raise ValueError("q shape %s, expected (%d, 300)" % (tuple(q.shape), B))
    out = torch.empty(B, 300, device=o.device, dtype=torch.float32)
    wq = bq = qh = None
    hkn = 0
    if next_packed is not None:
        wq, bq, hkn = next_packed
        qh = torch.empty(B, hkn, device=o.device, dtype=torch.float32)
    L = _lib.lib()
    _launch("mgnns_mha_tail_fwd", ("mgnns_mha_tail_fwd",), L.mgnns_mha_tail_fwd, _p(o), HK, _p(q), B, 300,
            _p(packed["fc_wp"]), _p(packed["fc_b"]), _p(packed["g1"]), _p(packed["be1"]), _p(packed["w1_wp"]),
            _p(packed["b1"]), _p(packed["w2_wp"]), _p(packed["b2"]), _p(packed["g2"]), _p(packed["be2"]), float(eps),
            _p(out), _p(wq), _p(bq), hkn, _p(qh), _stream())
    return out, qh


def pack_weight_bf16_split(w):
    """[N, K] fp32 -> (hi, lo) MFMA-fragment-major bf16 buffers (w ~ hi + lo) for mha_tail_bf16."""
    _chk(w, "weight", ndim=2)
    L = _lib.lib()
    n = L.mgnns_packed_bf16_weight_bytes(w.shape[0], w.shape[1])
    hi = torch.empty(n, dtype=torch.uint8, device=w.device)
    lo = torch.empty(n, dtype=torch.uint8, device=w.device)
    _lib.check(L.mgnns_pack_weight_bf16_split(_p(w), w.shape[0], w.shape[1], _p(hi), _p(lo), _stream()),
               "mgnns_pack_weight_bf16_split")
    return hi, lo


TAIL_BF16_KSPLIT = os.environ.get("MGNNS_TAIL_BF16_KSPLIT", "1") == "1"


def mha_tail_bf16(o, q, packed, eps, next_packed=None, terms=3, cluster=0, ksplit=None, next_linear=None):
    """bf16-MFMA fused tail.  packed: dict with fc, w1, w2 = (hi, lo) buffers and fc_b, g1, be1, b1, b2, g2, be2;
    next_packed = ((hi, lo), bq, HK_next) or None.  ksplit (default on): a tile's cluster of workgroups splits the K
    of fc and exchanges partial sums through scratch owned by `packed` (one per capture epoch and launch stream), the next
    layer's w_qs runs as a second launch (terms == 1: mha_proj_c16; terms == 3: the exact-fp32 GEMM on next_linear = (weight
    [HK_next, 300], bias), which the K-split form needs in place of next_packed); cluster: workgroups per 16-sample tile
    (0 = the library's default)."""
    import ctypes
    _chk(o, "o", ndim=2)
    _chk(q, "q", ndim=2)
    B, HK = o.shape
    if q.shape != (B, 300):
        raise ValueError("q shape %s, expected (%d, 300)" % (tuple(q.shape), B))
    out = torch.empty(B, 300, device=o.device, dtype=torch.float32)
    ptrs = [packed["fc"][0].data_ptr(), packed["fc"][1].data_ptr(), packed["w1"][0].data_ptr(),
            packed["w1"][1].data_ptr(), packed["w2"][0].data_ptr(), packed["w2"][1].data_ptr(), None, None]
    bq = qh = None
    hkn = 0
    use_ks = (TAIL_BF16_KSPLIT if ksplit is None else ksplit) and B > 0 and cluster != 1
    if int(terms) == 3 and use_ks and next_packed is not None and next_linear is None:
        use_ks = False                    # (the split-bf16 K split has no packed projection: without next_linear, the one-launch form)
    split_proj = int(terms) == 3 and use_ks and next_linear is not None
    if next_packed is not None and not split_proj:
        (wh, wl), bq, hkn = next_packed
        ptrs[6], ptrs[7] = wh.data_ptr(), wl.data_ptr()
        qh = torch.empty(B, hkn, device=o.device, dtype=torch.float32)
    arr = (ctypes.c_void_p * 8)(*ptrs)
    L = _lib.lib()
    scratch = counters = None
    if use_ks and int(terms) in (1, 3):
        tiles = (B + 15) // 16
        slot = packed.setdefault("_cluster_ws_ks", {})
        key = _scratch_key()                                     # per (capture epoch, launch stream), like the c16 tail's
        ws = slot.get(key)
        if ws is None or ws[0] < tiles or ws[1].device != o.device:
            if ws is not None:
                packed.setdefault("_retired", []).append(ws)     # a captured hipGraph may still hold its address: never freed
            ws = (tiles, torch.empty(L.mgnns_mha_tail_c16_scratch_floats(16 * tiles, 8), device=o.device, dtype=torch.float32),
                  zeros_i32(2 * tiles, o.device))
            _scratch_slot_put(slot, key, ws)
        scratch, counters = ws[1], ws[2]
    _launch("mgnns_mha_tail_bf16_fwd", ("mgnns_mha_tail_bf16_fwd",), L.mgnns_mha_tail_bf16_fwd, _p(o), HK, _p(q), B, 300,
            int(terms), arr, _p(packed["fc_b"]), _p(packed["g1"]), _p(packed["be1"]), _p(packed["b1"]), _p(packed["b2"]),
            _p(packed["g2"]), _p(packed["be2"]), float(eps), _p(out), _p(bq), hkn, _p(qh), int(cluster), _p(scratch), _p(counters),
            _stream())
    if split_proj:
        qh = linear(out, next_linear[0], next_linear[1])
    return out, qh


TAIL_C16_KSPLIT = os.environ.get("MGNNS_TAIL_C16_KSPLIT", "1") == "1"


def mha_tail_c16(c, q, packed, eps, next_packed=None, cluster=0, ksplit=None):
    """The bf16 fused tail behind sq_mha_folded_bf16 (mgnns_mha_tail_c16_fwd): c bf16 [B, H*300 rounded up to 32];
    packed["fc"] = the composed map fc . blockdiag(W_v); next_packed = ((hi, lo), bias, H*300) of the next layer's composed query
    map or None.  cluster: workgroups per 16-sample tile (0 = the library's default); ksplit (default on): the ranks split the K of
    the first product and exchange partial sums through scratch owned by `packed`, one per (capture epoch, launch stream)."""
    import ctypes
    _chk(c, "c", torch.bfloat16, 2)
    _chk(q, "q", ndim=2)
    B, HC = c.shape
    if q.shape != (B, 300):
        raise ValueError("q shape %s, expected (%d, 300)" % (tuple(q.shape), B))
    out = torch.empty(B, 300, device=c.device, dtype=torch.float32)
    ptrs = [packed["fc"][0].data_ptr(), None, packed["w1"][0].data_ptr(), None, packed["w2"][0].data_ptr(), None, None, None]
    bq = un = None
    hcn = 0
    if next_packed is not None:
        (wh, _wl), bq, hcn = next_packed
        ptrs[6] = wh.data_ptr()
        un = torch.empty(B, hcn, device=c.device, dtype=torch.float32)
    arr = (ctypes.c_void_p * 8)(*ptrs)
    L = _lib.lib()
    scratch = counters = None
    if (TAIL_C16_KSPLIT if ksplit is None else ksplit) and B > 0 and cluster != 1:
        tiles = (B + 15) // 16
        slot = packed.setdefault("_cluster_ws", {})
        key = _scratch_key()                                     # per (capture epoch, launch stream), like the channel tail's
        ws = slot.get(key)
        if ws is None or ws[0] < tiles or ws[1].device != c.device:
            if ws is not None:
                packed.setdefault("_retired", []).append(ws)     # a captured hipGraph may still hold its address: never freed
            ws = (tiles, torch.empty(L.mgnns_mha_tail_c16_scratch_floats(16 * tiles, 8), device=c.device, dtype=torch.float32),
                  zeros_i32(2 * tiles, c.device))
            _scratch_slot_put(slot, key, ws)
        scratch, counters = ws[1], ws[2]
    _launch("mgnns_mha_tail_c16_fwd", ("mgnns_mha_tail_c16_fwd",), L.mgnns_mha_tail_c16_fwd, _p(c), HC, _p(q), B, 300,
            arr, _p(packed["fc_b"]), _p(packed["g1"]), _p(packed["be1"]), _p(packed["b1"]), _p(packed["b2"]),
            _p(packed["g2"]), _p(packed["be2"]), float(eps), _p(out), _p(bq), hcn, _p(un), int(cluster), _p(scratch), _p(counters),
            _stream())
    return out, un


def layernorm(x, gamma, beta, eps=1e-6):
    D = x.shape[-1]
    x2 = _chk(x.reshape(-1, D), "x")
    _chk(gamma, "gamma", ndim=1)
    _chk(beta, "beta", ndim=1)
    y = torch.empty_like(x2)
    L = _lib.lib()
    _lib.check(L.mgnns_layernorm_fwd(_p(x2), x2.shape[0], D, _p(gamma), _p(beta), float(eps), _p(y), _stream()),
               "mgnns_layernorm_fwd")
    return y.view(x.shape)


# ---- dense bf16 GEMM (large graphs / large X.W) -------------------------------------------------------------------------
def _kpad(k):
    return (k + 63) // 64 * 64


def transpose_cast_bf16(x):
    """[K, N] fp32 -> bf16 [N, Kp] (K-contiguous, zero padded to a multiple of 64): the Bt operand of gemm_bf16_nt."""
    _chk(x, "x", ndim=2)
    K, N = x.shape
    y = torch.empty(N, _kpad(K), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().mgnns_transpose_cast_bf16(_p(x), K, N, y.shape[1], _p(y), _stream()), "mgnns_transpose_cast_bf16")
    return y


GEMM_BF16_KSPLIT = os.environ.get("MGNNS_GEMM_BF16_KSPLIT", "1") == "1"
_gemm_bf16_ws = {}


def _gemm_bf16_workspace(device):
    """Partial-sum slots + arrival counters of the dense bf16 GEMM's last-round K split: one buffer per (device, capture epoch,
    stream) -- launches on one stream are ordered, a captured graph may replay next to anything.  The K split is a two-launch scheme
    (partial sums, then a fix-up launch that reads exactly the slots the first wrote): no counters, nothing to zero, so the buffer
    is plain torch.empty (64 MB from the zero pool would be a fresh torch.zeros + stream synchronise per request)."""
    key = (str(device),) + _scratch_key()
    ws = _gemm_bf16_ws.get(key)
    if ws is None:
        ws = torch.empty(_lib.lib().mgnns_gemm_bf16_workspace_bytes(), dtype=torch.uint8, device=device)
        _gemm_bf16_ws[key] = ws
        if key[1]:
            _EPOCH_SLOTS.setdefault(key[1], []).append((_gemm_bf16_ws, key))
    return ws


def gemm_bf16_nt(a_bf16, bt_bf16, bias=None, act=ACT_NONE, n_valid=None, out=None, out_dtype=torch.float32, transposed_out=False):
    """act(A . Bt^T + bias): A bf16 [M, Kp], Bt bf16 [N, Kp] (Kp % 64 == 0) -> fp32 or bf16 [M, N].
    out: optional preallocated result; it may be a column slice [M, N] of a wider row-major matrix (row stride = its stride(0)),
    e.g. the K-padded operand of the next product.
    transposed_out (round 6): the result is stored as its transpose [N, M] (`out`, if given, is that [N, M] matrix or a column slice of a
    wider one); bias stays per column n of the product.  For products whose natural orientation has a small M: run the transpose
    (M = the long side) and keep the layout the consumer needs.  Needs ceil(M / 160) >= 8, N >= 256, K >= 320."""
    _chk(a_bf16, "A", torch.bfloat16, 2)
    _chk(bt_bf16, "Bt", torch.bfloat16, 2)
    M, Kp = a_bf16.shape
    N = bt_bf16.shape[0]
    if bt_bf16.shape[1] != Kp or Kp % 64:
        raise ValueError("A %s / Bt %s: K rows must match and be a multiple of 64" % (tuple(a_bf16.shape), tuple(bt_bf16.shape)))
    if bias is not None:
        _chk(bias, "bias", ndim=1)
    shape = (N, M) if transposed_out else (M, N)
    if out is None:
        c = torch.empty(*shape, device=a_bf16.device, dtype=out_dtype)
    else:
        c = out
        if not (torch.is_tensor(c) and c.is_cuda and c.dim() == 2 and tuple(c.shape) == shape and c.stride(1) == 1
                and c.dtype in (torch.float32, torch.bfloat16)):
            raise ValueError("out must be a [%d, %d] fp32 / bf16 device matrix with unit column stride" % shape)
    if transposed_out and ((M + 159) // 160 < 8 or N < 256 or Kp < 320):
        raise ValueError("transposed_out needs ceil(M / 160) >= 8, N >= 256, K >= 320 (M=%d N=%d K=%d)" % (M, N, Kp))
    ws = _gemm_bf16_workspace(a_bf16.device) if (GEMM_BF16_KSPLIT and M * N >= (1 << 20)) else None
    _lib.check(_lib.lib().mgnns_gemm_bf16_nt_fwd(_p(a_bf16), _p(bt_bf16), M, N, Kp, _p(bias), _p(c), c.stride(0),
                                                 (1 if c.dtype == torch.bfloat16 else 0) | (2 if transposed_out else 0), act, _p(ws),
                                                 0 if ws is None else ws.numel(),
                                                 _stream()), "mgnns_gemm_bf16_nt_fwd")
    return c


def gemm_bf16_set_form(form):
    """Tile shape of gemm_bf16_nt (tests / timings): -1 environment (MGNNS_GEMM_160, default by estimate), 0 round 4's kernels only,
    1 the 160 x 256 kernel whenever the shape fits it, 2 by the launcher's estimate, 3 the 320 x 256 kernel whenever it fits."""
    _lib.check(_lib.lib().mgnns_gemm_bf16_set_form(int(form)), "mgnns_gemm_bf16_set_form")


def dense_adj_matmul_bf16(adj_bf16, support, act=ACT_NONE):
    """act(adj @ support) with the DENSE adjacency kept in bf16 ([C, Kp], from cast_pad_bf16(adj, ld=Kp)) and the fp32
    support [C, F] transposed + cast on the fly: the dense counterpart of spmm_csr for graphs that are not sparse."""
    return gemm_bf16_nt(adj_bf16, transpose_cast_bf16(support), None, act)


# ---- f4: ResNet trunks (NHWC bf16 activations) ------------------------------------------------------------------
STEM_LD = 160      # stem weight rows: K = 3*7*7 = 147 zero padded to 5 MFMA k-steps


def conv_fold_bn(weight, conv_bias=None, bn=None, stem=False):
    """Fold an eval-mode BatchNorm (tuple gamma, beta, running_mean, running_var, eps -- or None) into a convolution
    weight [Cout, Cin, KH, KW] fp32 -> (bf16 [Cout, ld] with k = (kh, kw, c), or (c, kh, kw) padded to 160 for the stem;
    fp32 bias [Cout])."""
    _chk(weight, "weight", ndim=4)
    Cout, Cin, KH, KW = weight.shape
    ld = STEM_LD if stem else KH * KW * Cin
    if stem and (Cin, KH, KW) != (3, 7, 7):
        raise ValueError("stem weight must be [Cout, 3, 7, 7], got %s" % (tuple(weight.shape),))
    g = b = m = v = None
    eps = 0.0
    if bn is not None:
        g, b, m, v, eps = bn
        for t, n in ((g, "bn.weight"), (b, "bn.bias"), (m, "bn.running_mean"), (v, "bn.running_var")):
            _chk(t, n, ndim=1)
            if t.shape[0] != Cout:
                raise ValueError("%s has %d entries for %d output channels" % (n, t.shape[0], Cout))
    if conv_bias is not None:
        _chk(conv_bias, "conv_bias", ndim=1)
    wt = torch.empty(Cout, ld, device=weight.device, dtype=torch.bfloat16)
    bias = torch.empty(Cout, device=weight.device, dtype=torch.float32)
    _lib.check(_lib.lib().mgnns_conv_fold_bn_bf16(_p(weight), _p(conv_bias), Cout, Cin, KH, KW, _p(g), _p(b), _p(m), _p(v),
                                                  float(eps), 1 if stem else 0, ld, _p(wt), _p(bias), _stream()),
               "mgnns_conv_fold_bn_bf16")
    return wt, bias


def stem_conv7(img, wt, bias):
    """relu(bn1(conv1(img))): img [B,3,H,W] fp32 NCHW -> [B, OH, OW, 64] bf16 NHWC."""
    _chk(img, "img", ndim=4)
    _chk(wt, "wt", torch.bfloat16, 2)
    _chk(bias, "bias", ndim=1)
    B, C, H, W = img.shape
    if C != 3 or tuple(wt.shape) != (64, STEM_LD) or bias.shape[0] != 64:
        raise ValueError("stem expects img [B,3,H,W], wt [64,%d], bias [64]; got %s %s %s"
                         % (STEM_LD, tuple(img.shape), tuple(wt.shape), tuple(bias.shape)))
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64, device=img.device, dtype=torch.bfloat16)
    L = _lib.lib()
    _launch("mgnns_stem_conv7_fwd", ("mgnns_stem_conv7_fwd",), L.mgnns_stem_conv7_fwd, _p(img), B, H, W, _p(wt), _p(bias),
            _p(y), _stream())
    return y


def maxpool3x3s2_nhwc(x):
    _chk(x, "x", torch.bfloat16, 4)
    B, H, W, C = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, device=x.device, dtype=torch.bfloat16)
    L = _lib.lib()
    _launch("mgnns_maxpool3x3s2_nhwc_fwd", ("mgnns_maxpool3x3s2_nhwc_fwd",), L.mgnns_maxpool3x3s2_nhwc_fwd, _p(x), B, H, W, C,
            _p(y), _stream())
    return y


def conv_bf16_nhwc(x, wt, bias, ksize, stride=1, pad=0, residual=None, relu=True, out_nchw_f32=False):
    """relu?(conv(x) + bias + residual?): x [B,H,W,Cin] bf16, wt [Cout, k*k*Cin] bf16 (conv_fold_bn) -> [B,OH,OW,Cout]
    bf16, or [B,Cout,OH,OW] fp32 when out_nchw_f32."""
    _chk(x, "x", torch.bfloat16, 4)
    _chk(wt, "wt", torch.bfloat16, 2)
    _chk(bias, "bias", ndim=1)
    B, H, W, Cin = x.shape
    Cout = wt.shape[0]
    if wt.shape[1] != ksize * ksize * Cin or bias.shape[0] != Cout:
        raise ValueError("wt %s / bias %s do not match a %dx%d convolution of %d channels"
                         % (tuple(wt.shape), tuple(bias.shape), ksize, ksize, Cin))
    OH = (H + 2 * pad - ksize) // stride + 1
    OW = (W + 2 * pad - ksize) // stride + 1
    if residual is not None:
        _chk(residual, "residual", torch.bfloat16, 4)
        if tuple(residual.shape) != (B, OH, OW, Cout):
            raise ValueError("residual %s != output %s" % (tuple(residual.shape), (B, OH, OW, Cout)))
    if out_nchw_f32:
        y = torch.empty(B, Cout, OH, OW, device=x.device, dtype=torch.float32)
    else:
        y = torch.empty(B, OH, OW, Cout, device=x.device, dtype=torch.bfloat16)
    L = _lib.lib()
    _launch("mgnns_conv_bf16_nhwc_fwd", ("mgnns_conv_bf16_nhwc_fwd", ksize, Cin, Cout, stride, OH), L.mgnns_conv_bf16_nhwc_fwd,
            _p(x), B, H, W, Cin, _p(wt), _p(bias), Cout, ksize, ksize, stride, pad, _p(residual), 1 if relu else 0,
            1 if out_nchw_f32 else 0, _p(y), _stream())
    return y


# ---- measurement aid: in-graph timestamps --------------------------------------------------------------------------
_timeline = None          # (slots tensor [uint64 as int64], names list) while tools/graph_timeline.py is recording


def timeline_begin(device, n=64):
    global _timeline
    _timeline = (torch.zeros(n, dtype=torch.int64, device=device), [])
    return _timeline


def timeline_end():
    global _timeline
    t, _timeline = _timeline, None
    return t


def stamp(name):
    """Record `name` at the current point of the current stream (no-op unless a timeline is being recorded)."""
    if _timeline is None:
        return
    slots, names = _timeline
    if len(names) >= slots.numel():
        raise RuntimeError("timeline full")
    names.append(name)
    _lib.check(_lib.lib().mgnns_debug_stamp(_p(slots), len(names) - 1, _stream()), "mgnns_debug_stamp")
