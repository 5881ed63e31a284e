"""Text-level GCN channel with the reference's module surface (models/Text_GCN.py:36-275).

`Model(...)` keeps the constructor signature and parameter names of the reference class
(node_hidden, seq_edge_w, Linear).  forward() hands the whole batch of token ids to one HIP
kernel that builds each document's n-gram graph, looks the PMI edge ids up in a CSR map,
gathers node/edge embeddings and does the max-aggregation + sum read-out + ReLU
(the reference does this with a Python loop per document and three DGL kernels).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .pmi import PmiCsr


class Model(nn.Module):
    def __init__(self, class_num, hidden_size_node, vocab, n_gram, drop_out, edges_num, edges_matrix,
                 max_length=100, trainable_edges=True, pmi=None, cuda=True, is_padding=True,
                 word2vec_file='glove/glove.6B.300d.txt'):
        super().__init__()
        self.is_cuda = cuda
        self.is_padding = is_padding
        self.vocab = vocab
        self.node_hidden = nn.Embedding(len(vocab), hidden_size_node)
        self.edges_num = edges_num
        if trainable_edges:
            self.seq_edge_w = nn.Embedding.from_pretrained(torch.ones(edges_num, 1), freeze=False)
        else:
            if pmi is None:
                raise ValueError("trainable_edges=False needs the PMI weights `pmi` [edges_num, 1]")
            self.seq_edge_w = nn.Embedding.from_pretrained(torch.as_tensor(pmi).float().reshape(edges_num, 1), freeze=False)
        self.hidden_size_node = hidden_size_node
        emb = self.load_word2vec(word2vec_file)
        if emb is not None:
            self.node_hidden.weight.data.copy_(torch.as_tensor(emb))
        self.node_hidden.weight.requires_grad = True
        self.len_vocab = len(vocab)
        self.ngram = n_gram
        self.d = dict(zip(self.vocab, range(len(self.vocab))))
        self.max_length = max_length
        # dense [V,V] matrix (reference), scipy sparse or PmiCsr: kept as CSR, uploaded on first use
        self.edges_matrix = PmiCsr.coerce(edges_matrix)
        if self.edges_matrix.n_rows != len(vocab):
            raise ValueError("edges_matrix has %d rows, vocab has %d words" % (self.edges_matrix.n_rows, len(vocab)))
        if self.edges_matrix.max_eid() >= edges_num:
            raise ValueError("edges_matrix refers to edge id %d but edges_num is %d" % (self.edges_matrix.max_eid(), edges_num))
        self.dropout = nn.Dropout(p=drop_out)
        self.activation = nn.ReLU()
        self.Linear = nn.Linear(hidden_size_node, class_num, bias=True)   # constructed, never applied (TGCN:273)

    def word2id(self, word):
        return self.d.get(word, self.d.get('UNK'))

    def load_word2vec(self, word2vec_file):
        """GloVe initialisation of node_hidden (Text_GCN.py:105-121).  Needs the third-party
        `word2vec` reader and the GloVe file; without them node_hidden keeps its random init
        (weights normally arrive through load_state_dict)."""
        if not os.path.exists(word2vec_file):
            return None
        try:
            import word2vec
        except ImportError:
            return None
        model = word2vec.load(word2vec_file)
        rows = []
        for word in self.vocab:
            try:
                rows.append(model[word])
            except KeyError:
                rows.append(model['the'])
        return np.array(rows)

    def forward(self, doc_ids, is_20ng=None):
        """doc_ids [B,T] int64 (0 = PAD) -> relu(sum_nodes(max-aggregated node states)) [B, hidden]."""
        if self.training:
            raise RuntimeError("Text_GCN.Model: eval-mode forward only on the HIP path; call .eval()")
        if doc_ids.dim() != 2:
            raise ValueError("doc_ids must be [B,T]")
        doc_ids = doc_ids.long().contiguous()
        pmi_dev = self.edges_matrix.device_arrays(doc_ids.device)
        return ops.textgcn(doc_ids, self.node_hidden.weight.detach(), self.seq_edge_w.weight.detach(),
                           pmi_dev, self.ngram, self.max_length)
