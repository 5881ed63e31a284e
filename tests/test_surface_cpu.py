"""CPU-side checks of the drop-in boundary: the C ABI exports what include/mgnns_hip.h declares,
the nn.Module exposes the reference's state_dict surface, and nothing computes without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from mgnns_amd import _lib, ops, synth
from mgnns_amd.pmi import PmiCsr
from tests import helpers as H
from tests.model_util import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mgnns_hip.h")).read()
    declared = set(re.findall(r"\b(mgnns_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("mgnns_stream_t")
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libmgnns_hip.so does not export %s" % name
    assert set(_lib.SIGNATURES) | set(_lib.SIZE_GETTERS) | {"mgnns_last_error", "mgnns_abi_version", "mgnns_source_fingerprint"} == declared
    assert L.mgnns_abi_version() == _lib.ABI_VERSION
    # the library says which sources it was built from; a build right before this test (the driver's build()) makes them the tree's
    built, tree, same = _lib.source_state()
    assert len(built) == 16 and same, "libmgnns_hip.so was built from sources %s, the tree holds %s: rebuild" % (built, tree)
    assert _lib.check_sources("test") == tree


def test_a_library_older_than_its_sources_is_refused(monkeypatch):
    """tools/write_profiles.py files a profile only under the sources the measured library was built from: _lib.check_sources raises
    when the tree's fingerprint differs from the one compiled into the library (here: the tree's is faked)."""
    from mgnns_amd import build
    assert _lib.check_sources("test") == build.source_fingerprint()
    monkeypatch.setattr(build, "source_fingerprint", lambda: "0" * 16)
    built, tree, same = _lib.source_state()
    assert tree == "0" * 16 and not same
    with pytest.raises(_lib.MgnnsLibraryError, match="rebuild"):
        _lib.check_sources("tools/write_profiles.py")


def test_state_dict_surface_equals_reference():
    cfg = synth.CONFIGS["tumemo_b64"]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    adj = H.load_golden("adjacency.npz")
    m = build_model(cfg, pmi, count, adj["object_t04_A"], adj["place_t03_A"], np.zeros((7, 300), np.float32))
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == H.surface()           # keys AND shapes, dead parameters included (strict loading)


def test_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.layernorm(torch.zeros(2, 4), torch.ones(4), torch.zeros(4))
    cfg = synth.CONFIGS["mvsa_single_b8"]
    pmi, count = synth.synth_pmi(cfg.V, seed=3)
    adj = H.load_golden("adjacency.npz")
    m = build_model(cfg, pmi, count, adj["object_t04_A"], adj["place_t03_A"], np.zeros((7, 300), np.float32))
    inp = synth.make_inputs(cfg, B=2, pmi=pmi)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    with pytest.raises(RuntimeError, match="GPU only"):
        m(t["text"], t["text_lens"], t["text_mask"], t["object_feature"], t["place_feature"], t["object_inp"],
          t["place_inp"])


def test_pmi_csr_matches_dense_lookup():
    rs = np.random.RandomState(0)
    V = 60
    dense = np.zeros((V, V), dtype=np.int64)
    k = 1
    for i in range(V):
        for j in range(V):
            if rs.uniform() < 0.1:
                dense[i, j] = k
                k += 1
    m = PmiCsr.from_dense(dense)
    for i in range(V):
        for j in range(V):
            assert m[i, j] == dense[i, j]
    assert m.nnz == k - 1 and m.max_eid() == k - 1
    import scipy.sparse as sp
    m2 = PmiCsr.coerce(sp.csr_matrix(dense))
    assert np.array_equal(m2.col, m.col) and np.array_equal(m2.eid, m.eid)


def test_attention_choice_and_the_precision_mode():
    """The default is the reference's explicit formulation in every precision mode; 'folded' is an explicit choice that survives a
    change of the precision mode; 'auto' = the folded attention in the bf16 modes and the explicit one in fp32 mode; every fusion
    layer carries the resolved choice; unknown names raise."""
    from mgnns_amd.fusion import MultiHeadAttention
    cfg = synth.CONFIGS["mvsa_single_b8"]
    pmi, count = synth.synth_pmi(cfg.V, seed=3)
    adj = H.load_golden("adjacency.npz")
    m = build_model(cfg, pmi, count, adj["object_t04_A"], adj["place_t03_A"], np.zeros((7, 300), np.float32))
    layers = [x for x in m.modules() if isinstance(x, MultiHeadAttention)]
    assert layers and m.attention_choice == "faithful" and m.precision == "fp32" and m.attention == "faithful"
    for prec in ("bf16", "bf16x3", "fp32"):
        m.set_precision(prec)
        assert m.attention == "faithful" and all(x.attention == "faithful" and x.precision == prec for x in layers), prec
    m.set_attention("auto")
    for prec, want in (("bf16", "folded"), ("bf16x3", "folded"), ("fp32", "faithful")):
        m.set_precision(prec)
        assert m.attention == want and all(x.attention == want for x in layers), prec
    m.set_attention("folded").set_precision("fp32")
    assert m.attention == "folded" and all(x.attention == "folded" for x in layers)
    with pytest.raises(ValueError):
        m.set_attention("flash")


def test_every_schedule_is_a_valid_plan_without_a_gpu():
    """forward_plan only builds closures: every schedule must run each of the 13 segments exactly once, after the segments
    it depends on, on one of the four stream keys; cross-stream dependencies are exactly the data dependencies that live
    on another stream; 'auto' resolves by batch size; unknown names raise."""
    cfg = synth.CONFIGS["mvsa_single_b8"]
    pmi, count = synth.synth_pmi(cfg.V, seed=3)
    adj = H.load_golden("adjacency.npz")
    m = build_model(cfg, pmi, count, adj["object_t04_A"], adj["place_t03_A"], np.zeros((7, 300), np.float32))
    inp = synth.make_inputs(cfg, B=2, pmi=pmi)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    args = (t["text"], t["text_lens"], t["text_mask"], t["object_feature"], t["place_feature"], t["object_inp"], t["place_inp"])
    assert m.schedule == "auto" and m.resolve_schedule(2) == "small" and m.resolve_schedule(64) == "channels2" and m.resolve_schedule(128) == "place_bank_first"
    m.precision = "bf16x3"                                            # (the mode's own default: 'channels2' at every batch, NOTES_r05 section 1)
    assert m.resolve_schedule(2) == m.resolve_schedule(64) == m.resolve_schedule(256) == "channels2"
    m.precision = "fp32"
    for name, sched in m.SCHEDULES.items():
        plan, ctx = m.forward_plan(*args, schedule=name)
        assert ctx == {}                                              # nothing ran
        seen = {}
        for seg, skey, cross, fn in plan:
            # (the classifier head is no segment of its own when the four stacks carry its shares: model.split_head)
            assert (callable(fn) or (fn is None and seg == "head" and m.split_head)) and skey in ("main", "s1", "s2", "s3"), (name, seg, skey)
            base = seg.split("+")[0]
            assert base == seg and base not in seen, (name, seg)
            for d in m.SEGMENT_DEPS[base]:
                assert d in seen, (name, seg, d)
            want = {d for d in m.SEGMENT_DEPS[base] if seen[d] != skey}
            assert want <= set(cross) and all(seen[d] != skey for d in cross), (name, seg, cross)
            seen[base] = skey
        assert set(seen) == set(m.SEGMENT_DEPS), name
        assert plan[-1][0] == "head" and plan[-1][1] == "main", name
    with pytest.raises(ValueError):
        m.forward_plan(*args, schedule="no-such-schedule")


def test_default_schedules_are_pinned():
    """resolve_schedule('auto') for B in {16, 32, 64, 128, 256} x the precision modes: the defaults were chosen from alternating-run
    A/Bs on one box (NOTES_r04 / r05, +-1.5 % noise) -- changing one is a measured decision that edits this table, not a side effect."""
    from mgnns_amd.model import default_schedule
    table = {
        "fp32":   {16: "small", 32: "small", 64: "channels2", 128: "place_bank_first", 256: "place_bank_first"},
        "bf16":   {16: "small", 32: "small", 64: "channels2", 128: "place_bank_first", 256: "place_bank_first"},
        "bf16x3": {16: "channels2", 32: "channels2", 64: "channels2", 128: "channels2", 256: "channels2"},
    }
    for prec, row in table.items():
        for B, want in row.items():
            assert default_schedule(B, prec) == want, (prec, B)
    cfg = synth.CONFIGS["mvsa_single_b8"]
    pmi, count = synth.synth_pmi(cfg.V, seed=3)
    adj = H.load_golden("adjacency.npz")
    m = build_model(cfg, pmi, count, adj["object_t04_A"], adj["place_t03_A"], np.zeros((7, 300), np.float32))
    for att in ("faithful", "folded"):
        m.set_attention(att)
        for prec, row in table.items():
            m.set_precision(prec)
            assert {B: m.resolve_schedule(B) for B in row} == row, (prec, att)
            assert all(m.resolve_schedule(B) in m.SCHEDULES for B in row)


def test_dense_gemm_tile_shape_is_chosen_by_host_arithmetic():
    """csrc/gemm_bf16.hip::mg_gemm_pick through mgnns_gemm_bf16_pick_form (no device call): the tile shape of the configs[4] products
    on a 256-CU device.  4 = 160 x 256 tiles (one round of 252 tiles for 10 000 x 1024), 5 = 320 x 256 (one round of 256 tiles for
    10 000 x 2048), 0 = round 4's kernels; the choices are the ones tools/dev/gemm_time.py measured as fastest (NOTES_r05 section 9)."""
    L = _lib.lib()
    pick = lambda M, N, Kp, ws=1, cu=256: L.mgnns_gemm_bf16_pick_form(M, N, Kp, ws, cu)
    assert pick(10000, 1024, 10048) == 4 and pick(10000, 1024, 10048, ws=0) == 4      # adjacency product, F = 1024
    assert pick(10000, 2048, 10048) == 5                                              # adjacency product, F = 2048
    assert pick(10000, 1024, 2048) == 4 and pick(10000, 2048, 1024) == 5              # X . W products
    assert pick(10000, 1024, 320) == 4
    assert pick(20154, 1200, 320) == 0            # the BiLSTM's folded table: short K, the 256 x 128 kernel's small fixed cost wins
    assert pick(512, 10000, 2048) == 0            # the read-out: fewer row blocks than XCDs (the column-split map of the 256 x 128 kernel)
    assert pick(8192, 2048, 4096) == 0 and pick(8192, 2048, 4096, ws=0) == 5          # exactly one round of 256 x 256 tiles, if its workspace is there
    assert pick(64, 256, 64) == 0
    assert pick(0, 256, 64) < 0 and pick(256, 256, 100) < 0                           # bad arguments are refused
