"""Host-side batch assembly in front of the forward: text -> padded ids / lens / mask exactly as the reference's
dataset does it (utils/Multi_GCN_Co_att_dataset.py:233-265: word2id with UNK, right-pad with PAD to the split's
max length, mask = ids != PAD), written into PINNED host buffers and copied asynchronously into the static device
buffers a captured forward reads (mgnns_amd.graph.GraphedForward) -- the reference builds per-sample LongTensors,
collates them in DataLoader workers and moves seven tensors with blocking .to(device) calls
(engine/Multi_GCN_Multihead_Att_engine.py:803-810).  No graph construction happens on the host: the text-GCN
kernel builds the n-gram graphs from the token ids on the device."""
import numpy as np
import torch

from . import _lib

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

    def encode_ids(self, rows):
        """The same from already tokenised texts (sequences of word ids, e.g. TokenCache rows): pure numpy padding."""
        n = len(rows)
        if n > self.B:
            raise ValueError("batch of %d texts exceeds the assembler's batch size %d" % (n, self.B))
        ids = self.text.numpy()
        ids[:] = self.w2i.pad
        lens = self.lens.numpy()
        lens[:] = 0
        if n:
            ln = np.fromiter((len(r) for r in rows), dtype=np.int64, count=n)
            if int(ln.max()) > self.T:
                raise ValueError("text of %d tokens does not fit max_len %d" % (int(ln.max()), self.T))
            lens[:n] = ln
            flat = np.concatenate([np.asarray(r, dtype=np.int64) for r in rows]) if int(ln.sum()) else np.zeros(0, np.int64)
            rowi = np.repeat(np.arange(n), ln)
            coli = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
            ids[rowi, coli] = flat
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


class TokenCache:
    """word2id applied ONCE per text (the reference re-tokenises every text at every __getitem__, DSET:252-255, spread over
    DataLoader worker processes): rows of int64 ids, what BatchAssembler.encode_ids pads."""

    def __init__(self, vocab, texts):
        w2i = Word2Id(vocab)
        self.rows = [np.fromiter((w2i(w) for w in t.split(' ')), dtype=np.int64) for t in texts]

    def __len__(self):
        return len(self.rows)

    def batch(self, lo, hi):
        return self.rows[lo:hi]


class PipelinedForward:
    """Host batch assembly, H2D copy and the captured forward as a two-deep pipeline:

        host      : assemble(i+1) into pinned set (i+1)%2 ........ assemble(i+2) ...
        copy strm : ....................... H2D(i+1) -> device staging set (i+1)%2
        main strm : [copy staging(i) -> graph inputs][replay(i)] .... [copy staging(i+1) -> ...][replay(i+1)]

    While the GPU replays batch i the host pads batch i+1 into the OTHER pinned buffer set and the copy stream moves it
    into the other device staging set; the only thing left on the compute stream between two replays is a device-to-device
    copy of the three small text tensors into the graph's static inputs (~2 us).  The reference does word2id / padding in
    DataLoader workers and seven blocking `.to(device)` calls per batch on the training thread (ENGINE:803-810).
    Reuse of a pinned set is safe once its H2D has completed (event), of a staging set once the replay that consumed it
    has been enqueued behind its D2D copy (stream order)."""

    def __init__(self, graphed, vocab, max_len, batch_size, device):
        self.gf = graphed
        self.asm = [BatchAssembler(vocab, max_len, batch_size, device) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.h2d_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]
        self.B = int(batch_size)

    def _stage(self, k, batch, from_ids):
        a = self.asm[k]
        self.h2d_done[k].synchronize()              # the pinned set's previous H2D has left the host buffers
        (a.encode_ids if from_ids else a.encode)(batch)
        self.copy_stream.wait_event(self.consumed[k])   # the staging set's previous contents were consumed
        a.to_device(self.copy_stream)
        self.h2d_done[k].record(self.copy_stream)

    def run(self, batches, from_ids=False, on_logits=None):
        """batches: iterable of lists of texts (or of id rows with from_ids=True), each of exactly batch_size entries.
        on_logits(i, logits) is called with the graph's static output after replay i has been ENQUEUED (the tensor is
        overwritten by the next replay: consume it on the stream, e.g. metrics.update).  Returns the number of batches."""
        main = torch.cuda.current_stream()
        it = iter(batches)
        try:
            nxt = next(it)
        except StopIteration:
            return 0
        self._stage(0, nxt, from_ids)
        i = 0
        while True:
            k = i & 1
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            main.wait_event(self.h2d_done[k])
            text, lens, mask = self.asm[k].dev
            self.gf.static_in[0].copy_(text, non_blocking=True)
            self.gf.static_in[1].copy_(lens, non_blocking=True)
            self.gf.static_in[2].copy_(mask, non_blocking=True)
            self.consumed[k].record(main)
            out = self.gf.replay()
            if nxt is not None:
                self._stage(k ^ 1, nxt, from_ids)    # overlaps the replay just enqueued
            if on_logits is not None:
                on_logits(i, out)
            i += 1
            if nxt is None:
                return i

    def run_in_flight(self, pipeline, batches, from_ids=False, on_logits=None):
        """run() on a graph.GraphedPipeline: batch i is launched on a capture of its own without joining the streams behind
        batch i-1 (two forwards in flight); on_logits(i-1, logits) runs -- on the caller's stream, behind a wait for that
        capture -- AFTER batch i has been launched, so the consumer never holds up the next launch."""
        main = torch.cuda.current_stream()
        it = iter(batches)
        try:
            nxt = next(it)
        except StopIteration:
            return 0
        self._stage(0, nxt, from_ids)
        i, pending = 0, None
        while True:
            k = i & 1
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            item = pipeline.items[pipeline.i]
            item.wait()                              # its previous replay may still be reading its static inputs
            main.wait_event(self.h2d_done[k])
            text, lens, mask = self.asm[k].dev
            item.static_in[0].copy_(text, non_blocking=True)
            item.static_in[1].copy_(lens, non_blocking=True)
            item.static_in[2].copy_(mask, non_blocking=True)
            self.consumed[k].record(main)
            ready = torch.cuda.Event()
            ready.record(main)
            pipeline.replay(ready)
            if nxt is not None:
                self._stage(k ^ 1, nxt, from_ids)    # overlaps the replay just enqueued
            if pending is not None and on_logits is not None:
                pending[1].wait()
                on_logits(pending[0], pending[1].static_out)
            pending = (i, item)
            i += 1
            if nxt is None:
                if on_logits is not None:
                    pending[1].wait()
                    on_logits(pending[0], pending[1].static_out)
                return i

    def run_serial(self, batches, from_ids=False, on_logits=None):
        """The same work without overlap (assemble -> blocking H2D -> replay -> wait), the reference's order."""
        i = 0
        a = self.asm[0]
        for b in batches:
            (a.encode_ids if from_ids else a.encode)(b)
            text, lens, mask = a.to_device()
            self.gf.static_in[0].copy_(text, non_blocking=True)
            self.gf.static_in[1].copy_(lens, non_blocking=True)
            self.gf.static_in[2].copy_(mask, non_blocking=True)
            out = self.gf.replay()
            torch.cuda.current_stream().synchronize()
            _lib.take_status()                       # a persistent launch of THIS replay that gave up a bounded wait raises here
            if on_logits is not None:
                on_logits(i, out)
            i += 1
        return i

    def finish(self):
        """Host-side end of a run() / run_in_flight(): wait for everything enqueued, then check the persistent launches' status
        word.  Every replay reports a bounded wait that ran out in an EARLIER replay (graph.GraphedForward.replay); this
        covers the last ones -- call it before results accumulated by on_logits (metrics, logits copies) are read on the host."""
        torch.cuda.synchronize()
        _lib.take_status()
