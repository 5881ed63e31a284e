"""Whole-forward hipGraphs: capture one forward once, replay it per batch.

The eval forward is static for a given batch shape -- no host-side decisions, no device-to-host syncs (the
reference's Text_GCN forward syncs and loops in Python per document, Text_GCN.py:232-234) -- so it is captured
with PyTorch's HIP-graph support: kernels launched through the C ABI go to the capturing stream like any other
launch.  Inputs live in static device buffers that the caller (or `copy_inputs`) fills before each replay.

Two capture modes:

* ``mode="segments"`` (default): the model's forward_plan() -- 13 segments, each a LINEAR chain of launches on one of
  four streams -- is captured as one linear hipGraph per segment; a replay launches them on the model's four streams
  with the plan's event waits in between.  Who runs concurrently with whom is then decided by four in-order HIP
  streams, exactly as in eager execution, at ~10 graph launches per forward instead of ~36 kernel launches (consecutive
  segments of one stream with nothing between them share a graph; the classifier head rides at the end of the four stacks).
* ``mode="single"``: the whole multi-stream forward as ONE graph with four parallel branches.  The graph runtime maps
  branches onto its own internal streams; on this stack it ran at most three of the four fusion-stack chains at a
  time (tools/graph_timeline.py: the fourth started when a sibling had finished, ~200 us late) and the result moved
  170-252 k samples/s with GPU_MAX_HW_QUEUES.  Kept for comparison.
"""
import os
import subprocess
import sys
import warnings

import torch

from . import _lib

_PROBED = {}


def single_graph_probe(schedule, timeout=900, precision="fp32", attention="faithful"):
    """Can the runtime capture the multi-stream forward of `schedule` as ONE graph?  hipStreamEndCapture has been seen to
    SEGFAULT on some stream topologies (ROCm 7.0 / 7.2) -- not an exception a process survives -- so the first one-graph
    capture of a topology is tried in a CHILD process on a small synthetic model (the topology, not the size, is what the
    runtime trips over).  Cached per process."""
    # the TOPOLOGY also depends on the mode: with a packing plan of the text mask (bf16 / bf16x3 + faithful) the image->text stacks
    # wait for the text-GCN segment as well (model.PLAN_SITE) -- the child captures the mode the caller is in
    key = schedule if (precision, attention) == ("fp32", "faithful") else (schedule, precision, attention)
    if key not in _PROBED:
        env = dict(os.environ, MGNNS_GRAPH_MODE="segments")
        try:
            r = subprocess.run([sys.executable, "-m", "mgnns_amd.graph", "--probe-single", schedule, precision, attention], env=env,
                               timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _PROBED[key] = r.returncode == 0
        except (subprocess.TimeoutExpired, OSError):
            _PROBED[key] = False
    return _PROBED[key]


class GraphedForward:
    def __init__(self, model, example_args, warmup=3, post=None, mode=None, _probe_child=False, settle=True):
        """example_args: the 7 forward arguments on the GPU (text_lens included, int64 on the device).
        post: optional callable applied to the logits INSIDE the capture (e.g. ShardedForward.gather: the RCCL
        all-gather becomes a node of the last graph)."""
        self.model = model
        self.pick_ms = None
        # default 'segments': with ~33 launches per forward it beat the one-graph form at every batch size measured in round 2
        # (B=256: 0.86 vs 0.96 ms, B=32: 0.49 vs 0.53), and the one-graph capture is where the runtime can segfault (below);
        # 'auto' still times both
        self.mode = mode or os.environ.get("MGNNS_GRAPH_MODE", "segments")
        if self.mode not in ("segments", "single", "auto"):
            raise ValueError("mode must be 'segments', 'single' or 'auto'")
        if self.mode == "auto" and post is not None:
            self.mode = "segments"                  # a collective inside: every rank must capture the same thing, once
        if not getattr(model, "use_streams", True):
            self.mode = "single"                    # one stream: the forward is one linear chain anyway
        elif self.mode in ("single", "auto") and hasattr(model, "resolve_schedule") and not _probe_child:
            # hipStreamEndCapture SEGFAULTS inside the runtime (ROCm 7.0 / 7.2) on the single-graph form of every schedule
            # tried but 'channels' (lgcn_side, banks_first, channels2: python -X faulthandler points at capture_end) -- not
            # an exception this process could survive: the one-graph form of a topology is first captured in a child process
            sched = model.resolve_schedule(example_args[0].shape[0])
            if not single_graph_probe(sched, precision=getattr(model, "precision", "fp32"), attention=getattr(model, "attention", "faithful")):
                warnings.warn("MGNNS_GRAPH_MODE=%s: the runtime cannot capture schedule %r as one graph (probe child failed); "
                              "using one graph per segment" % (self.mode, sched))
                self.mode = "segments"
        # a fresh scratch epoch: the persistent launches captured below get scratch buffers of their own (ops._scratch_key),
        # so this graph can be replayed next to any other forward of the same model; the model keeps superseded weight packs
        # alive while a graph that may hold their addresses exists
        from . import ops as _ops
        self._epoch = _ops.new_scratch_epoch()
        prev_epoch = _ops.set_scratch_epoch(self._epoch)
        model._live_graphs = getattr(model, "_live_graphs", 0) + 1
        _ops.capture_born()                         # layer-level packs (fusion.py) park instead of freeing while this lives
        try:
            self._build(model, example_args, warmup, post, settle)
        finally:
            _ops.set_scratch_epoch(prev_epoch)

    def __del__(self):
        try:
            if getattr(self, "_epoch", None) is not None:
                from . import ops as _ops
                _ops.release_scratch_epoch(self._epoch)     # the persistent launches' scratch slots created for this capture
                _ops.capture_gone()
                self._epoch = None
            m = getattr(self, "model", None)
            if m is not None and getattr(m, "_live_graphs", 0) > 0:
                m._live_graphs -= 1
                if m._live_graphs == 0 and hasattr(m, "_wt_retired"):
                    m._wt_retired.clear()
        except Exception:          # interpreter shutdown: torch's module machinery may already be torn down
            pass

    def _build(self, model, example_args, warmup, post, settle):
        self.static_in = [a.clone() if torch.is_tensor(a) else a for a in example_args]
        for a in self.static_in:
            if torch.is_tensor(a) and not a.is_cuda:
                raise RuntimeError("GraphedForward needs every tensor argument on the GPU (text_lens too)")
        if getattr(model, "use_streams", True):
            model._side_streams(self.static_in[0].device)   # bound to hardware queues other than the CALLER's stream's
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                 # builds weight caches / workspaces / function attributes
                self.static_out = model(*self.static_in)
                if post is not None:
                    self.static_out = post(self.static_out)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from . import ops as _ops
        _ops.reserve_zeros(self.static_in[0].device)      # counters of the launches captured below: no fill nodes in the graphs
        if self.mode in ("single", "auto"):
            try:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph), torch.no_grad():
                    self._single_out = model(*self.static_in)
                    if post is not None:
                        self._single_out = post(self._single_out)
                self.static_out = self._single_out
            except Exception:
                if self.mode == "single":
                    raise
                self.graph, self.mode = None, "segments"          # the runtime refused this topology as one graph
                torch.cuda.synchronize()
        if self.mode in ("segments", "auto"):
            self._capture_segments(post)
        if self.mode == "auto":
            # Both forms compute the same thing; which is faster depends on the batch (11 graph launches per forward cost
            # more than they gain below ~100 samples): time a few replays of each, keep the faster, drop the other.
            self.mode = self._pick_mode()
            if self.mode == "single":
                self._segs, self._ctx = None, None
                self.static_out = self._single_out
            else:
                self.graph = None
        # The first ~20 replays of a fresh capture run 2-3 % slower than the steady state (B=256: 0.861 ms per forward over
        # replays 6-35, 0.840 over replays 31-130; 'auto' used to look faster than 'segments' only because its mode timing
        # had already replayed 16 times).  Settle here, once, so a short measurement sees the steady state.
        # (settle=False: the caller settles later -- with a collective captured inside, every rank must first agree that all
        # of them captured it, or the ranks' collective sequences diverge: bench.py)
        if settle:
            self.settle()

    def settle(self):
        for _ in range(int(os.environ.get("MGNNS_GRAPH_SETTLE", "16"))):
            self.replay()
        torch.cuda.synchronize()

    def _pick_mode(self, reps=6):
        import time
        best = {}
        for m in ("single", "segments"):
            self.mode = m
            for _ in range(2):
                self.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                self.replay()
            torch.cuda.synchronize()
            best[m] = time.perf_counter() - t0
        self.pick_ms = {m: round(t / reps * 1e3, 4) for m, t in best.items()}      # reported by bench.py
        return min(best, key=best.get)

    # ---- one linear graph per plan segment ------------------------------------------------------------------------
    def _capture_segments(self, post):
        model = self.model
        dev = self.static_in[0].device
        self._side = dict(model._side_streams(dev))
        # (a `post` step -- the sharded forward's all-gather -- is captured behind the head: keep the head a segment then)
        plan, ctx = model.forward_plan(*self.static_in, split_head=None if post is None else False)
        plan = [seg for seg in plan if seg[3] is not None]
        # Consecutive segments of ONE stream become one graph when nothing has to happen between them: the later one waits
        # for no other stream, and no other stream waits for the earlier one (an event cannot be recorded inside a graph).
        # E.g. a channel's label GCN + memory bank: one graph launch and one boundary less on the bank -> tail -> stack chain.
        merged = []
        for name, skey, deps, fn in plan:
            prev = next((m for m in reversed(merged) if m[1] == skey), None)
            prev_needed_elsewhere = prev is not None and any(prev[0][-1] in d and k != skey for _, k, d, _ in plan)
            if prev is not None and not deps and not prev_needed_elsewhere and merged[-1] is prev:
                prev[0].append(name)
                prev[3].append(fn)
            else:
                merged.append(([name], skey, deps, [fn]))
        plan = [(names[-1], skey, deps, (lambda fs=fns: [f() for f in fs])) for names, skey, deps, fns in merged]
        self.segment_names = [names for names, _, _, _ in merged]
        ctx.prepare()                               # ordinary memory, before any capture (the split head's buffers)
        self._ctx = ctx                             # keeps every cross-segment tensor (graph outputs) alive
        needed = {d for _, _, deps, _ in plan for d in deps}
        self._segs = []
        # a dedicated capture stream per plan stream: per-stream workspaces (ops._gemm_workspace) are keyed by the
        # launch stream, and segments that replay concurrently must not share one
        self._cap = {k: torch.cuda.Stream(device=dev) for k in ["main"] + list(self._side)}
        from . import ops
        for st in self._cap.values():               # their workspaces exist BEFORE any capture (ordinary memory)
            with torch.cuda.stream(st):
                ops._gemm_workspace(dev)
        with torch.no_grad():
            for i, (name, skey, deps, fn) in enumerate(plan):
                g = torch.cuda.CUDAGraph()
                last = i == len(plan) - 1
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=self._cap[skey]):
                    fn()
                    if last and post is not None:
                        ctx['logits'] = post(ctx['logits'])
                ev = torch.cuda.Event() if name in needed else None
                self._segs.append((name, skey, deps, g, ev))
        torch.cuda.synchronize()
        self._seg_out = self.static_out = ctx['logits']
        self._fork = torch.cuda.Event()

    def copy_inputs(self, *args):
        for dst, src in zip(self.static_in, args):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)

    def replay(self):
        """Run the captured forward on the static inputs; returns the static logits tensor."""
        # A replay runs no C-ABI launcher, so nothing else would ever look at the persistent launches' status word: a bounded
        # wait that ran out inside an EARLIER replay is reported here (host read of a pinned word, no synchronisation);
        # result() is the form that covers the replay it returns.
        _lib.take_status()
        if self.mode == "single":
            self.graph.replay()
            return self._single_out
        main = torch.cuda.current_stream()
        streams = dict(self._side, main=main)
        self._fork.record(main)                     # inputs (copy_inputs) and the previous replay's readers are on `main`
        for st in self._side.values():
            st.wait_event(self._fork)
        done = {}
        for name, skey, deps, g, ev in self._segs:
            st = streams[skey]
            for d in deps:
                if done[d][1] is not st:
                    st.wait_event(done[d][0])
            with torch.cuda.stream(st):
                g.replay()
                if ev is not None:
                    ev.record(st)
                    done[name] = (ev, st)
        for st in self._side.values():
            main.wait_stream(st)
        if hasattr(self, "_end"):                   # a later replay_async() orders its streams behind THIS replay as well
            for skey, st in streams.items():
                self._end[skey].record(main)
        return self._seg_out

    def replay_async(self, inputs_ready=None):
        """Replay WITHOUT the end-of-forward join (segments mode): the caller's stream is not made to wait for the side streams,
        so the next replay of ANOTHER GraphedForward (buffers of its own) can start on a stream as soon as that stream's own
        work of this one is done -- two forwards in flight (GraphedPipeline).  Each stream first waits for the end of THIS
        instance's previous replay on the other streams (its buffers are reused) and for `inputs_ready` (an event behind
        copy_inputs; None: the static inputs are unchanged).  Returns the static logits; wait() before reading them."""
        if self.mode == "single":
            raise RuntimeError("replay_async needs one graph per segment (mode 'segments')")
        _lib.take_status()                          # an earlier replay's bounded wait that ran out (see replay())
        main = torch.cuda.current_stream()
        streams = dict(self._side, main=main)
        if not hasattr(self, "_end"):
            self._end = {k: torch.cuda.Event() for k in streams}
        for skey, st in streams.items():
            for k2, ev in self._end.items():
                if k2 != skey:
                    st.wait_event(ev)
            if inputs_ready is not None:
                st.wait_event(inputs_ready)
        done = {}
        for name, skey, deps, g, ev in self._segs:
            st = streams[skey]
            for d in deps:
                if done[d][1] is not st:
                    st.wait_event(done[d][0])
            with torch.cuda.stream(st):
                g.replay()
                if ev is not None:
                    ev.record(st)
                    done[name] = (ev, st)
        for skey, st in streams.items():
            self._end[skey].record(st)
        return self._seg_out

    def wait(self):
        """Make the caller's stream wait for the last replay_async()."""
        main = torch.cuda.current_stream()
        for ev in getattr(self, "_end", {}).values():
            main.wait_event(ev)

    def result(self):
        """The logits of the last replay()/replay_async() for the HOST: waits for it, then checks the persistent launches'
        status word -- a launch inside the replayed graphs that gave up a bounded wait raises here instead of handing back
        invalid logits."""
        self.wait()
        torch.cuda.current_stream().synchronize()
        _lib.take_status()
        return self.static_out

    def __call__(self, *args):
        self.copy_inputs(*args)
        return self.replay()


class GraphedPipeline:
    """`depth` captures of the same forward (buffers of their own, the model's four streams shared), replayed round robin
    without a join between them: while the fusion stacks of one batch still run on three streams, the BiLSTM chain / memory
    banks of the next start on the streams that are done -- the seam between two forwards (join, fork, first graph launch:
    ~33 us of a 0.76-ms forward in profiles/r03_timeline.txt) and the idle compute units at both ends of a forward are filled
    with the neighbouring batch's work.  Throughput, not latency: every forward still does all of its work on its own batch."""

    def __init__(self, model, example_args, depth=2, **kw):
        self.items = [GraphedForward(model, example_args, mode="segments", **kw) for _ in range(depth)]
        self.i = 0

    @classmethod
    def of(cls, items):
        """From existing captures (mode 'segments') of the same forward."""
        if any(it.mode != "segments" for it in items):
            raise ValueError("GraphedPipeline needs captures with one graph per segment")
        self = cls.__new__(cls)
        self.items, self.i = list(items), 0
        return self

    def copy_inputs(self, *args):
        """New inputs for the NEXT replay (on the caller's stream); returns the event the replay must wait for."""
        self.items[self.i].wait()                   # its previous replay may still be reading the static inputs
        self.items[self.i].copy_inputs(*args)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def replay(self, inputs_ready=None):
        """Start the next forward; returns its GraphedForward (item.wait() orders the caller's stream behind it, its
        static_out holds the logits)."""
        it = self.items[self.i]
        self.i = (self.i + 1) % len(self.items)
        it.replay_async(inputs_ready)
        return it

    def wait(self):
        for it in self.items:
            it.wait()



def _probe_single_main(schedule, precision="fp32", attention="faithful"):
    """Child of single_graph_probe: one-graph capture of `schedule` on a small synthetic model; exit code 0 iff it captured,
    replayed and reproduced the eager logits."""
    from . import harness, synth
    cfg = synth.CONFIGS["tumemo_b64"]
    dev = "cuda:0"
    pmi, count = synth.synth_pmi(cfg.V, seed=2)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp = synth.make_inputs(cfg, B=8, seed=3, pmi=pmi)
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
    model.schedule = schedule
    model.set_precision(precision).set_attention(attention)
    args = list(harness.call_args(inp, dev))
    args[1] = args[1].to(dev)
    with torch.no_grad():
        ref = model(*args).clone()
    g = GraphedForward(model, args, mode="single", _probe_child=True)
    out = g.replay().clone()
    torch.cuda.synchronize()
    return 0 if g.mode == "single" and torch.allclose(out, ref, atol=1e-5 if precision == "fp32" else 1e-2) else 1


if __name__ == "__main__":
    if len(sys.argv) in (3, 5) and sys.argv[1] == "--probe-single":
        sys.exit(_probe_single_main(*sys.argv[2:]))
    sys.exit(2)
