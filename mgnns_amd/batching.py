"""Host-side batch assembly in front of the forward: text -> padded ids / lens / mask exactly as the reference's
dataset does it (utils/Multi_GCN_Co_att_dataset.py:233-265: word2id with UNK, right-pad with PAD to the split's
max length, mask = ids != PAD), written into PINNED host buffers and copied asynchronously into the static device
buffers a captured forward reads (mgnns_amd.graph.GraphedForward) -- the reference builds per-sample LongTensors,
collates them in DataLoader workers and moves seven tensors with blocking .to(device) calls
(engine/Multi_GCN_Multihead_Att_engine.py:803-810).  No graph construction happens on the host: the text-GCN
kernel builds the n-gram graphs from the token ids on the device."""
import numpy as np
import torch

from .vocab import Word2Id


class BatchAssembler:
    def __init__(self, vocab, max_len, batch_size, device=None):
        self.w2i = Word2Id(vocab)
        self.T = int(max_len)
        self.B = int(batch_size)
        pin = device is not None and torch.cuda.is_available()
        self.text = torch.zeros(self.B, self.T, dtype=torch.int64, pin_memory=pin)
        self.lens = torch.zeros(self.B, dtype=torch.int64, pin_memory=pin)
        self.mask = torch.zeros(self.B, self.T, dtype=torch.float32, pin_memory=pin)
        self.device = device
        self.dev = None
        if device is not None:
            self.dev = tuple(torch.empty_like(t, device=device) for t in (self.text, self.lens, self.mask))

    def encode(self, texts):
        """Fill the host buffers from up to B whitespace-tokenised strings; returns (text, lens, mask) host views.
        A text longer than max_len raises, as the reference's fixed-size assignment would (DSET:241)."""
        n = len(texts)
        if n > self.B:
            raise ValueError("batch of %d texts exceeds the assembler's batch size %d" % (n, self.B))
        ids = self.text.numpy()
        ids[:] = self.w2i.pad
        lens = self.lens.numpy()
        lens[:] = 0
        for b, t in enumerate(texts):
            row = [self.w2i(w) for w in t.split(' ')]
            if len(row) > self.T:
                raise ValueError("text of %d tokens does not fit max_len %d" % (len(row), self.T))
            ids[b, :len(row)] = row
            lens[b] = len(row)
        self.mask.numpy()[:] = (ids != self.w2i.pad)
        return self.text[:n], self.lens[:n], self.mask[:n]

    def to_device(self, stream=None):
        """Asynchronous H2D of the three text tensors into the static device buffers."""
        if self.dev is None:
            raise RuntimeError("BatchAssembler was built without a device")
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for d, h in zip(self.dev, (self.text, self.lens, self.mask)):
                d.copy_(h, non_blocking=True)
        return self.dev
