"""Host-side rows either side of the path: vocabulary, sparse PMI edge-map builder
and batch assembly against goldens produced by the reference's own build_vocab / cal_PMI / dataset padding
(oracle/gen_goldens.py::gold_hostside) on a slice of the shipped val split.  CPU only, bit-exact."""
import os
import tempfile

import numpy as np
import torch

from mgnns_amd.batching import BatchAssembler
from mgnns_amd.pmi import PmiCsr, build_pmi, load_pmi, save_pmi
from mgnns_amd.vocab import Word2Id, build_vocab
from tests import helpers as H


def _g():
    return H.load_golden("hostside.npz")


def test_build_vocab_equals_reference():
    g = _g()
    vocab = build_vocab([str(t) for t in g["texts"]], 2)
    assert vocab == [str(w) for w in g["vocab"]]
    assert vocab[0] == "PAD" and vocab[1] == "UNK"


def test_sparse_pmi_builder_equals_reference_cal_PMI():
    g = _g()
    texts = [str(t) for t in g["texts"]]
    vocab = [str(w) for w in g["vocab"]]
    weights, pmi, count = build_pmi(texts, vocab, window_size=5, min_cooccurence=2)
    assert count == int(g["pmi_count"])
    ref = PmiCsr.from_coo(g["pmi_rows"], g["pmi_cols"], g["pmi_eids"], len(vocab))
    assert np.array_equal(pmi.row_ptr, ref.row_ptr)
    assert np.array_equal(pmi.col, ref.col)
    assert np.array_equal(pmi.eid, ref.eid)                       # same ids: they index the learned seq_edge_w
    assert np.array_equal(weights, g["pmi_weights"])              # same float32 PMI values
    assert pmi[0, 0] == 0 and pmi.row_ptr[1] == 0                 # PAD is never a source
    # round trip through the compact on-disk form
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "pmi.npz")
        save_pmi(path, weights, pmi, count)
        w2, p2, c2 = load_pmi(path)
        assert c2 == count and np.array_equal(w2, weights) and np.array_equal(p2.col, pmi.col) and np.array_equal(p2.eid, pmi.eid)


def test_batch_assembler_equals_reference_padding():
    g = _g()
    vocab = [str(w) for w in g["vocab"]]
    texts = [str(t) for t in g["pad_texts"]]
    T = g["pad_ids"].shape[1]
    asm = BatchAssembler(vocab, max_len=T, batch_size=len(texts) + 3)
    text, lens, mask = asm.encode(texts)
    assert np.array_equal(text.numpy(), g["pad_ids"])
    assert np.array_equal(lens.numpy(), g["pad_lens"])
    assert np.array_equal(mask.numpy(), g["pad_mask"])
    assert text.dtype == torch.int64 and mask.dtype == torch.float32
    w2i = Word2Id(vocab)
    assert w2i("zzzz_unknown_word") == 1                           # UNK
    # unused tail rows of the fixed-size buffers are all PAD / length 0 / mask 0
    assert int(asm.text[len(texts):].abs().sum()) == 0 and int(asm.lens[len(texts):].sum()) == 0


def test_metric_scores_from_confusion_match_sklearn():
    """mgnns_amd.metrics.scores_from_confusion == accuracy_score / f1_score(micro, macro, weighted) as the engine
    computes them (ENGINE:833-838), including classes absent from the targets or from the predictions."""
    from sklearn.metrics import accuracy_score, f1_score
    from mgnns_amd.metrics import scores_from_confusion
    rs = np.random.RandomState(0)
    for NL in (3, 7):
        for trial in range(6):
            y = rs.randint(0, NL, size=257)
            p = rs.randint(0, NL if trial else NL - 1, size=257)       # trial 0: the last class is never predicted
            if trial == 3:
                y[y == NL - 1] = 0                                       # ... and here never a target
            if trial == 4:
                p = y.copy()                                             # perfect
            conf = np.zeros((NL, NL), np.int64)
            np.add.at(conf, (y, p), 1)
            s = scores_from_confusion(conf)
            ref = {"acc": accuracy_score(y, p), "micro_f1": f1_score(y, p, average="micro"),
                   "macro_f1": f1_score(y, p, average="macro"), "weighted_f1": f1_score(y, p, average="weighted")}
            for k, v in ref.items():
                assert abs(s[k] - v) < 1e-12, (NL, trial, k)


def test_token_cache_and_encode_ids_equal_string_path():
    """Tokenise-once path (TokenCache rows -> BatchAssembler.encode_ids, vectorised padding) == the per-batch string path
    == the reference's padding golden."""
    from mgnns_amd.batching import TokenCache
    g = _g()
    vocab = [str(w) for w in g["vocab"]]
    texts = [str(t) for t in g["pad_texts"]]
    T = g["pad_ids"].shape[1]
    a = BatchAssembler(vocab, max_len=T, batch_size=len(texts) + 3)
    b = BatchAssembler(vocab, max_len=T, batch_size=len(texts) + 3)
    a.encode(texts)
    tc = TokenCache(vocab, texts)
    text, lens, mask = b.encode_ids(tc.batch(0, len(texts)))
    assert torch.equal(a.text, b.text) and torch.equal(a.lens, b.lens) and torch.equal(a.mask, b.mask)
    assert np.array_equal(text.numpy(), g["pad_ids"]) and np.array_equal(lens.numpy(), g["pad_lens"])
    b.encode_ids(tc.batch(0, 2))                      # a shorter batch leaves the tail rows PAD / 0
    assert int(b.text[2:].abs().sum()) == 0 and int(b.lens[2:].sum()) == 0 and float(b.mask[2:].sum()) == 0.0
    import pytest
    with pytest.raises(ValueError, match="max_len"):
        b.encode_ids([np.arange(1, T + 2)])
