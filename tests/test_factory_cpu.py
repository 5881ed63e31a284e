"""The reference's factory surface (models/Multi_GCN_Multihead_att.py:586-642) on a temporary data root: the keyword
call of Tumblr_Multi_GCN_Multihead_Att.py:144-157 must construct the model unchanged -- vocabulary and PMI edge map
from <data_root>/all_anno_json/train_all_anno.json, both CNN trunks from their checkpoints, adjacency pickles relative
to the working directory.  Construction only (no GPU); the forward through it is tests/test_model_gpu.py."""
import inspect
import json
import os
import pickle

import numpy as np
import pytest
import torch

from mgnns_amd import model as M
from mgnns_amd import trunk
from mgnns_amd.pmi import build_pmi
from tests import helpers as H

# the reference's parameter lists, in order (MODEL:598-599, 619-627)
REF_TEXT_MODEL = ["data_root_path", "vocab_root_path", "text_min_count", "window_size", "num_labels", "ngram",
                  "text_dropout", "min_cooccurence"]
REF_FACTORY = ["opt", "num_labels", "object_num_classes", "place_num_classes", "object_t", "place_t", "data_root_path",
               "vocab_root_path", "text_min_count", "window_size", "ngram", "min_cooccurence", "text_dropout", "pretrained",
               "object_adj_file", "place_adj_file", "in_channel"]


def make_data_root(root, with_weights=True):
    """<root>/data/{all_anno_json/train_all_anno.json, adj/*.pkl, glove/tumblr_label_glove.pkl} + <root>/weights/*."""
    g = H.load_golden("hostside.npz")
    adj = H.load_golden("adjacency.npz")
    d = os.path.join(root, "data")
    os.makedirs(os.path.join(d, "all_anno_json"))
    os.makedirs(os.path.join(d, "adj"))
    os.makedirs(os.path.join(d, "glove"))
    with open(os.path.join(d, "all_anno_json", "train_all_anno.json"), "w") as f:
        for i, t in enumerate(g["texts"]):
            f.write(json.dumps({"id": i, "text": str(t), "label": "happy"}) + "\n")
    for tag, name in (("object", "tumblr_objects_adj.pkl"), ("place", "tumblr_resnet50_places_adj.pkl")):
        with open(os.path.join(d, "adj", name), "wb") as f:
            pickle.dump({"adj": adj[tag + "_counts"], "nums": adj[tag + "_nums"]}, f)
    with open(os.path.join(d, "glove", "tumblr_label_glove.pkl"), "wb") as f:
        pickle.dump(np.random.RandomState(0).standard_normal((7, 300)), f)         # float64, as the shipped pickle
    if with_weights:
        os.makedirs(os.path.join(root, "weights"))
        torch.manual_seed(0)
        torch.save(trunk.resnet101().state_dict(), os.path.join(root, "weights", "resnet101-5d3b4d8f.pth"))
        sd = {"module." + k: v for k, v in trunk.resnet50(365).state_dict().items()}       # DataParallel prefixes (MODEL:593)
        torch.save({"arch": "resnet50", "state_dict": sd}, os.path.join(root, "weights", "resnet50_places365.pth.tar"))
    return g, adj


def opt_for(vocab_size):
    # Tumblr_Multi_GCN_Multihead_Att.py:100-112 with the script's defaults
    return {'emb_path': None, 'bidirectional': True, 'hidden_size': 150, 'emb_size': 300, 'num_layers': 2, 'dropout': 0.5,
            'emb_type': 'random', 'vocab_size': vocab_size, 'stack_num': 2, 'n_head': 4, 'd_kv': 128, 'is_regu': False}


def test_factory_signatures_are_the_references():
    assert list(inspect.signature(M.Text_model).parameters) == REF_TEXT_MODEL
    sig = inspect.signature(M.multi_gcn_multihead_att_model)
    assert list(sig.parameters) == REF_FACTORY
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert d == {"text_dropout": 0.5, "pretrained": True, "object_adj_file": None, "place_adj_file": None, "in_channel": 300}
    assert list(inspect.signature(M.place_resnet).parameters) == ["arch"]


def test_factory_with_the_training_scripts_literal_kwargs(tmp_path, monkeypatch):
    g, adj = make_data_root(str(tmp_path))
    monkeypatch.chdir(tmp_path)                    # the script's paths are relative to its working directory
    from mgnns_amd.vocab import get_vocab_list
    vocab = get_vocab_list('data', 'data', 2)      # MAIN:95-97 (also writes data/vocab/vocab-2.txt like the reference)
    assert vocab == [str(w) for w in g["vocab"]]
    assert os.path.exists(os.path.join("data", "vocab", "vocab-2.txt"))
    opt = opt_for(len(vocab))
    model = M.multi_gcn_multihead_att_model(opt=opt,
                                            num_labels=7,
                                            object_num_classes=80, place_num_classes=365,
                                            object_t=0.4, place_t=0.3,
                                            data_root_path='data', vocab_root_path='data',
                                            text_min_count=2,
                                            window_size=5,
                                            ngram=4,
                                            min_cooccurence=2,
                                            text_dropout=0.5,
                                            pretrained=True,
                                            object_adj_file='data/adj/tumblr_objects_adj.pkl',
                                            place_adj_file='data/adj/tumblr_resnet50_places_adj.pkl',
                                            in_channel=300)
    assert isinstance(model, M.Multi_GCN_Multihead_Att)
    # text channel: the edge map and edge count of the reference's cal_PMI on these texts (golden from the reference)
    tf = model.text_features
    assert tf.vocab == vocab and tf.edges_num == int(g["pmi_count"])
    assert np.array_equal(tf.edges_matrix.eid, g["pmi_eids"].astype(np.int32))
    assert tuple(tf.seq_edge_w.weight.shape) == (int(g["pmi_count"]), 1)
    assert float(tf.seq_edge_w.weight.detach().min()) == 1.0            # trainable_edges=True: initialised to ones (TGCN:67-69)
    # adjacency parameters through gen_A at the script's thresholds
    assert np.array_equal(model.object_A.detach().numpy(), adj["object_t04_A"].astype(np.float32))
    assert np.array_equal(model.place_A.detach().numpy(), adj["place_t03_A"].astype(np.float32))
    # trunks: 8-entry Sequential with the checkpoint's values, 'module.' stripped
    ck = torch.load(os.path.join("weights", "resnet50_places365.pth.tar"))["state_dict"]
    assert torch.equal(model.place_features[0].weight, ck["module.conv1.weight"])
    assert torch.equal(model.place_features[7][2].conv3.weight, ck["module.layer4.2.conv3.weight"])
    ck = torch.load(os.path.join("weights", "resnet101-5d3b4d8f.pth"))
    assert torch.equal(model.object_features[6][22].conv2.weight, ck["layer3.22.conv2.weight"])
    assert len(model.object_features) == 8 and len(model.place_features) == 8
    # the label query was read from the module-level pickle path (MODEL:19-27)
    assert tuple(model.label_query.shape) == (7, 300) and model.label_query.dtype == torch.float32
    # what the script and the engine do next (MAIN:164, ENGINE:274-293)
    groups = model.get_config_optim(5e-5, 0.1)
    assert len(groups) == 12 and groups[1]['lr'] == pytest.approx(5e-6)
    assert model.image_normalization_mean == [0.485, 0.456, 0.406]
    # the state_dict surface outside the trunks and the vocabulary-sized tables is the reference's
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = H.surface()
    for k, s in ref.items():
        if k in ("embedding.weight", "text_features.node_hidden.weight"):
            assert mine[k] == (len(vocab), 300)
        elif k == "text_features.seq_edge_w.weight":
            assert mine[k] == (int(g["pmi_count"]), 1)
        else:
            assert mine[k] == s, k
    assert sum(k.startswith("object_features.") for k in mine) > 500


def test_factory_missing_checkpoints_fail_loudly(tmp_path, monkeypatch):
    make_data_root(str(tmp_path), with_weights=False)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "no_hub"))
    with pytest.raises(FileNotFoundError, match="resnet101"):
        M.object_resnet(pretrained=True)
    with pytest.raises(FileNotFoundError, match="places365"):
        M.place_resnet()
    monkeypatch.setenv("MGNNS_TRUNK_INIT", "random")         # documented escape: weights arrive by load_state_dict later
    assert isinstance(M.place_resnet(), trunk.ResNet)
    assert isinstance(M.object_resnet(pretrained=True), trunk.ResNet)


def test_text_model_equals_from_parts(tmp_path, monkeypatch):
    g, _ = make_data_root(str(tmp_path), with_weights=False)
    monkeypatch.chdir(tmp_path)
    tm = M.Text_model('data', 'data', 2, 5, 7, 4, 0.5, 2)
    vocab = [str(w) for w in g["vocab"]]
    w, pmi, count = build_pmi([str(t) for t in g["texts"]], vocab, window_size=5, min_cooccurence=2)
    tm2 = M.Text_model_from_parts(vocab, pmi, count, 7, 4, 0.5, edges_weights=w)
    assert tm.edges_num == tm2.edges_num == count
    assert np.array_equal(tm.edges_matrix.col, tm2.edges_matrix.col)
    assert {k: tuple(v.shape) for k, v in tm.state_dict().items()} == {k: tuple(v.shape) for k, v in tm2.state_dict().items()}


def test_checkpoint_loader_refuses_code_executing_pickles(tmp_path, monkeypatch):
    """MODEL:589 loads a third-party .pth.tar: numpy scalars next to the tensors (Places365's 'best_prec1') load through the
    safe unpickler; a pickle that needs the full one is refused unless MGNNS_TRUST_CHECKPOINTS=1."""
    import numpy as np
    from mgnns_amd.model import _load_checkpoint
    p = str(tmp_path / "a.pth.tar")
    torch.save({"arch": "resnet50", "best_prec1": np.float64(54.7), "state_dict": {"w": torch.ones(3)}}, p)
    ck = _load_checkpoint(p)
    assert float(ck["best_prec1"]) == 54.7 and torch.equal(ck["state_dict"]["w"], torch.ones(3))
    hit = []

    class Evil:
        def __reduce__(self):
            return (hit.append, ("executed",))

    torch.save({"x": Evil()}, p)
    monkeypatch.delenv("MGNNS_TRUST_CHECKPOINTS", raising=False)
    with pytest.raises(RuntimeError, match="MGNNS_TRUST_CHECKPOINTS"):
        _load_checkpoint(p)
    assert not hit
    monkeypatch.setenv("MGNNS_TRUST_CHECKPOINTS", "1")
    with pytest.warns(UserWarning, match="code-executing"):
        assert "x" in _load_checkpoint(p)          # the explicit opt-in loads it (the reduce call ran on a copy of `hit`)
