"""Generate tests/golden/*.npz by running the UNMODIFIED reference classes on CPU.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference).
    python oracle/gen_goldens.py            # writes tests/golden/
Inputs are seeded (mgnns_amd.synth); parameters are filled by NAME
(synth.param_value) so fixtures hold seeds + outputs, not weights.  Each fixture is
also checked here against oracle/restatement.py so a drift shows up at generation.
"""
import json
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from mgnns_amd import synth                      # noqa: E402
from oracle import golden_inputs as GI, ref_shims, restatement as R   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = ref_shims.REF


def fill_module(mod, prefix="", skip=()):
    """Fill every parameter/buffer of a reference module by state_dict name."""
    sd = mod.state_dict()
    new = {}
    for k, v in sd.items():
        if k in skip:
            new[k] = v
        else:
            new[k] = torch.from_numpy(synth.param_value(prefix + k, tuple(v.shape)))
    mod.load_state_dict(new)
    return {prefix + k: v.clone() for k, v in new.items()}


def maxdiff(a, b):
    return float((torch.as_tensor(a) - torch.as_tensor(b)).abs().max())


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("  wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------
def gold_adjacency(ns):
    print("[adjacency] gen_A / gen_adj on the shipped co-occurrence pickles")
    out = {}
    for tag, f, C in (("object", "data/adj/tumblr_objects_adj.pkl", 80),
                      ("place", "data/adj/tumblr_resnet50_places_adj.pkl", 365)):
        path = os.path.join(REF, f)
        raw = pickle.load(open(path, "rb"))
        out[tag + "_counts"] = np.asarray(raw["adj"]).astype(np.float32)
        out[tag + "_nums"] = np.asarray(raw["nums"]).astype(np.float32)
        assert np.array_equal(out[tag + "_counts"].astype(np.float64), np.asarray(raw["adj"]))
        assert np.array_equal(out[tag + "_nums"].astype(np.float64), np.asarray(raw["nums"]))
        for t in (0.3, 0.4, 0.5, 0.6):
            A64, _ = ns.UTIL.gen_A(C, t, path, 0.2)
            A = torch.from_numpy(A64).float()             # MODEL:341
            adj = ns.UTIL.gen_adj(A)
            key = "%s_t%02d" % (tag, int(round(t * 10)))
            out[key + "_A"] = A.numpy()
            out[key + "_adj"] = adj.numpy()
            mine_A = torch.from_numpy(R.gen_A(raw["adj"], raw["nums"], t, 0.2)).float()
            assert maxdiff(mine_A, A) == 0.0, key
            assert maxdiff(R.gen_adj(A), adj) == 0.0, key
            print("   %s: nnz(A)=%d" % (key, int((A != 0).sum())))
    save("adjacency.npz", **out)
    return out


def gold_image_gcn(ns, adjs):
    print("[image gcn] GraphConvolution x2 + LeakyReLU + read-out")
    gc1 = ns.MODEL.GraphConvolution(300, 1024)
    gc2 = ns.MODEL.GraphConvolution(1024, 2048)
    p = {}
    p.update(fill_module(gc1, "gc1."))
    p.update(fill_module(gc2, "gc2."))
    lrelu = torch.nn.LeakyReLU(0.2)
    proj = torch.from_numpy(GI.gcn_projection())
    out = {}
    for tag, C, key in (("object", 80, "object_t04"), ("place", 365, "place_t03")):
        A = torch.from_numpy(adjs[key + "_A"])
        X, pooled = (torch.from_numpy(a) for a in GI.image_gcn_case(tag))
        with torch.no_grad():
            adj = ns.UTIL.gen_adj(A).detach()
            G = gc2(lrelu(gc1(X, adj)), adj)              # MODEL:470-472
            x = torch.matmul(pooled, G.transpose(0, 1))   # MODEL:473-474
        mine = R.image_gcn(A, X, p["gc1.weight"], p["gc2.weight"])
        assert maxdiff(mine, G) < 1e-5, maxdiff(mine, G)
        out[tag + "_Gproj"] = G @ proj
        out[tag + "_x"] = x
        out[tag + "_Gabsmax"] = G.abs().max()
    save("image_gcn.npz", **out)


def gold_label_attention(ns):
    print("[label attention] Attention(300, C, 5 heads)")
    label = pickle.load(open(os.path.join(REF, "data/tumblr_label_glove.pkl"), "rb"))
    label = torch.from_numpy(np.array(label))             # f64, MODEL:26-27
    out = {"label_query": label.float()}
    for tag, C in (("object", 80), ("place", 365)):
        att = ns.MODEL.Attention(hid_dim=300, image_dim=C, n_heads=5, dropout=0.5).eval()
        p = fill_module(att, tag + "_attention.")
        lin5 = torch.nn.Linear(300, 100)
        xlin = torch.nn.Linear(700, 300)
        p.update(fill_module(lin5, tag + "_linear_5."))
        p.update(fill_module(xlin, tag + "_x_linear."))
        key = torch.from_numpy(GI.label_attention_key(tag))
        with torch.no_grad():
            y = att(query=label, key=key, value=key)      # MODEL:476
            z = xlin(lin5(y).view(5, -1))                 # MODEL:477-479
        mine = R.label_attention(p, tag + "_attention", label, key)
        assert maxdiff(mine, y) < 1e-5
        assert maxdiff(R.label_attention_tail(p, tag, mine), z) < 1e-5
        out[tag + "_y"] = y
        out[tag + "_z"] = z
    save("label_attention.npz", **out)


def gold_layernorm(ns):
    print("[layernorm] custom LayerNorm (unbiased std, eps on std)")
    ln = ns.SUBM.LayerNorm(300)
    p = fill_module(ln, "ln.")
    rs = np.random.RandomState(5)
    x = torch.from_numpy((2.0 * rs.standard_normal((9, 300)) + 0.5).astype(np.float32))
    x[3] = x[3, 0]                                        # constant row: std = 0 -> eps path
    with torch.no_grad():
        y = ln(x)
    assert maxdiff(R.layer_norm(x, p["ln.gamma"], p["ln.beta"]), y) == 0.0
    save("layernorm.npz", x=x, y=y)


def gold_mha(ns):
    print("[mha] MyMultiHeadAttention single-query layers")
    out = {}
    for H, tag, L, masked in GI.MHA_CASES:
        name = "h%d_%s" % (H, tag)
        layer = ns.MOUD.MyMultiHeadAttention(H, 300, 128, dropout=0.5, need_mask=masked).eval()
        p = fill_module(layer, name + ".")
        q, bank, mask = (None if a is None else torch.from_numpy(a) for a in GI.mha_case(H, tag, L, masked))
        with torch.no_grad():
            o, attn = layer(q=q, k=bank, v=bank, mask=mask)
        mo, ma = R.sq_mha_layer(p, name, q, bank, mask, H, 128)
        assert maxdiff(mo, o) < 2e-5 and maxdiff(ma, attn) < 1e-5, (maxdiff(mo, o), maxdiff(ma, attn))
        out[name + "_out"] = o
        out[name + "_attn"] = attn
        if H > 1:
            # is_regu=True (submodules.py:38-52, 84-93; moudles.py:220-229): the same layer with the head-difference term
            regu = ns.MOUD.MyMultiHeadAttention(H, 300, 128, dropout=0.5, need_mask=masked, is_regu=True).eval()
            regu.load_state_dict(layer.state_dict())
            with torch.no_grad():
                o2, attn2, hd = regu(q=q, k=bank, v=bank, mask=mask)
            assert maxdiff(o2, o) == 0.0 and maxdiff(attn2, attn) == 0.0
            assert maxdiff(R.sq_mha_layer(p, name, q, bank, mask, H, 128, return_head_diff=True)[2], hd) < 1e-6
            out[name + "_head_diff"] = hd
    save("mha.npz", **out)


def make_vocab(V):
    return ["PAD", "UNK"] + ["w%d" % i for i in range(2, V)]


def gold_text_gcn(ns):
    print("[text gcn] Text_GCN.Model.forward on the DGL stand-in")
    out = {}
    V, B, T = 3000, 12, 40
    pmi, count = synth.synth_pmi(V, per_row=8, seed=31)
    for ngram in (1, 4):
        tok, lens, _ = synth.synth_tokens(B, T, V, pmi, seed=100 + ngram)
        tok[2, 3] = 0                                     # an interior PAD: must be skipped (TGCN:147-150)
        tok[5, :] = 0
        tok[5, 0] = 17                                    # single-token doc
        m = ns.TGCN.Model(class_num=7, hidden_size_node=300, vocab=make_vocab(V), n_gram=ngram,
                          drop_out=0.5, edges_num=count, edges_matrix=pmi,
                          pmi=torch.ones(count, 1), cuda=True, trainable_edges=True).eval()
        p = fill_module(m, "text_features.")
        with torch.no_grad():
            y = m(torch.from_numpy(tok))
        mine = R.text_gcn(tok, p["text_features.node_hidden.weight"],
                          p["text_features.seq_edge_w.weight"], pmi, ngram)
        d = maxdiff(mine, y)
        print("   ngram=%d max|restatement-ref|=%.2e (|out|max %.2f; summation order only)"
              % (ngram, d, float(y.abs().max())))
        assert d < 1e-5 * float(y.abs().max()) + 1e-6, d
        out["ng%d_tok" % ngram] = tok
        out["ng%d_out" % ngram] = y
    out["V"] = V
    out["pmi_seed"] = 31
    out["count"] = count
    save("text_gcn.npz", **out)


def build_full(ns, cfg, num_labels, pmi, count, object_t=0.4, place_t=0.3):
    vocab = make_vocab(cfg.V)
    text_model = ns.TGCN.Model(class_num=num_labels, hidden_size_node=300, vocab=vocab, n_gram=cfg.ngram,
                               drop_out=0.5, edges_num=count, edges_matrix=pmi,
                               pmi=torch.ones(count, 1), cuda=True, trainable_edges=True)
    model = ns.MODEL.Multi_GCN_Multihead_Att(
        cfg.opt(), num_labels, text_model=text_model,
        object_model=ref_shims.PresetTrunk(), place_model=ref_shims.PresetTrunk(),
        object_num_classes=80, place_num_classes=365, object_t=object_t, place_t=place_t,
        in_channel=300,
        object_adj_file=os.path.join(REF, "data/adj/tumblr_objects_adj.pkl"),
        place_adj_file=os.path.join(REF, "data/adj/tumblr_resnet50_places_adj.pkl")).eval()
    return model


def gold_full(ns):
    print("[full forward] Multi_GCN_Multihead_Att.forward, identity trunks")
    surface = None
    for cfg_name, B in (("mvsa_single_b8", 8), ("tumemo_b64", 6), ("mvsa_multiple_b256", 4)):
        cfg = synth.CONFIGS[cfg_name]
        pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
        model = build_full(ns, cfg, cfg.NL, pmi, count)
        p = fill_module(model, skip=("object_A", "place_A"))
        if cfg_name == "tumemo_b64":
            surface = {k: list(v.shape) for k, v in model.state_dict().items()}
        inp = synth.make_inputs(cfg, B=B, pmi=pmi)
        ti = {k: torch.from_numpy(v) for k, v in inp.items()}
        with torch.no_grad():
            logits = model(ti["text"], ti["text_lens"], ti["text_mask"], ti["object_feature"],
                           ti["place_feature"], ti["object_inp"], ti["place_inp"])
        lq = ns.MODEL.glove_label_embedding.float()
        mine, parts = R.forward(p, ti, pmi, cfg.n_head, cfg.d_kv, cfg.stack_num, cfg.ngram,
                                label_query=lq, return_parts=True)
        d = maxdiff(mine, logits)
        print("   %s B=%d: max|restatement-ref| on logits = %.2e (|logits|max %.3f)"
              % (cfg_name, B, d, float(logits.abs().max())))
        assert d < 2e-5, d
        save("full_%s.npz" % cfg_name, B=B, logits=logits, label_query=lq,
             text=inp["text"], text_lens=inp["text_lens"],
             text_feature=parts["text_feature"], object_x=parts["object_x"], place_x=parts["place_x"],
             object_att=parts["object_att"], place_att=parts["place_att"],
             tio=parts["tio"], iot=parts["iot"])
    with open(os.path.join(OUT, "state_dict_surface.json"), "w") as f:
        json.dump(surface, f, indent=0, sort_keys=True)
    print("  wrote state_dict_surface.json (%d keys)" % len(surface))


def gold_text_bank(ns):
    print("[text bank] get_text_memory_bank (embedding + packed BiLSTM)")
    cfg = synth.Config("bank", B=6, T=20, V=500, n_head=1, stack_num=1)
    pmi, count = synth.synth_pmi(cfg.V, seed=3)
    model = build_full(ns, cfg, 7, pmi, count)
    p = fill_module(model, skip=("object_A", "place_A"))
    tok, lens, _ = synth.synth_tokens(cfg.B, cfg.T, cfg.V, pmi, seed=8)
    with torch.no_grad():
        bank, last = model.get_text_memory_bank(torch.from_numpy(tok), torch.from_numpy(lens))
    mine = R.text_memory_bank(p, torch.from_numpy(tok), torch.from_numpy(lens))
    assert maxdiff(mine, bank) < 1e-6
    save("text_bank.npz", tok=tok, lens=lens, bank=bank, V=cfg.V)


def gold_hostside(ns):
    """Reference host-side data prep on a small slice of the shipped val split: build_vocab + cal_PMI
    (utils/vocab_new.py, utils/pmi.py) and the dataset's word2id/_padding (utils/Multi_GCN_Co_att_dataset.py)."""
    print("[host side] vocab, PMI edge map, batch padding")
    import importlib
    import tempfile
    texts = []
    with open(os.path.join(REF, "data/all_anno_json/val_all_anno.json")) as f:
        for line in f:
            texts.append(json.loads(line)["text"])
            if len(texts) == 300:
                break
    texts.append(" ".join("w%d" % (i % 7) for i in range(100)))      # >= 100 tokens: dropped by cal_PMI's padding
    texts.append(" ".join(texts[0].split(" ")[:6]))
    root = tempfile.mkdtemp(prefix="mgnns_pmi_")
    os.makedirs(os.path.join(root, "all_anno_json"))
    os.makedirs(os.path.join(root, "vocab"))
    with open(os.path.join(root, "all_anno_json", "train_all_anno.json"), "w") as f:
        for t in texts:
            f.write(json.dumps({"text": t}) + "\n")
    VOC = importlib.import_module("utils.vocab_new")
    PMI = importlib.import_module("utils.pmi")
    vocab = VOC.get_vocab_list(root, root, 2)                       # builds + writes vocab-2.txt
    weights, mappings, count = PMI.cal_PMI(root, root, min_count=2, phase="train", window_size=5, min_cooccurence=2)
    r, c = np.nonzero(mappings)
    print("   V=%d, edges=%d" % (len(vocab), count - 1))
    # dataset padding
    DS = importlib.import_module("utils.Multi_GCN_Co_att_dataset")
    ds = DS.Tumblr_Dataset.__new__(DS.Tumblr_Dataset)
    ds.vocab = vocab
    ds.d = dict(zip(vocab, range(len(vocab))))
    ds.pad_idx = ds.d["PAD"]
    ds.text_max_length = 40
    sample = [t for t in texts[:40] if len(t.split(" ")) <= 40][:12] + ["zzzz_unknown_word " + texts[1].split(" ")[0]]
    ids, lens, masks = [], [], []
    for t in sample:
        content = list(map(lambda x: ds.word2id(x), t.split(" ")))
        cp, n, m = ds._padding(content)
        ids.append(cp.numpy())
        lens.append(n)
        masks.append(m.numpy())
    save("hostside.npz", texts=np.array(texts), vocab=np.array(vocab), pmi_rows=r, pmi_cols=c, pmi_eids=mappings[r, c],
         pmi_weights=weights.numpy(), pmi_count=count, pad_texts=np.array(sample), pad_ids=np.stack(ids),
         pad_lens=np.array(lens), pad_mask=np.stack(masks))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ns = ref_shims.install()
    adjs = gold_adjacency(ns)
    gold_image_gcn(ns, adjs)
    gold_label_attention(ns)
    gold_layernorm(ns)
    gold_mha(ns)
    gold_text_gcn(ns)
    gold_text_bank(ns)
    gold_full(ns)
    gold_hostside(ns)
    print("done")


if __name__ == "__main__":
    main()
