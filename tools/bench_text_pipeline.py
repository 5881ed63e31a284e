#!/usr/bin/env python
"""SURVEY 8 row f3, measured: real texts (the 302-text val-split fixture, cycled) -> vocabulary ids -> padded pinned buffers
-> H2D -> captured forward, serial vs two-deep pipelined, from strings and from a tokenise-once cache.  Feature maps stay
resident (they come from the on-device trunks in a real pipeline).  Prints one JSON object.

    python tools/bench_text_pipeline.py [--batches 60] [--batch 256]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(dev="cuda:0", n_batches=60, B=256, dtype="bf16"):
    import numpy as np
    import torch
    from mgnns_amd import harness, synth
    from mgnns_amd.batching import PipelinedForward, TokenCache
    from mgnns_amd.graph import GraphedForward, GraphedPipeline
    from mgnns_amd.pmi import build_pmi
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hostside.npz")
    g = np.load(golden)
    texts = [str(t) for t in g["texts"] if len(str(t).split(" ")) <= 100]
    vocab = [str(w) for w in g["vocab"]]
    weights, pmi, count = build_pmi(texts, vocab, window_size=5, min_cooccurence=2)
    cfg = synth.Config("realtext", B=B, T=100, V=len(vocab), NL=3, n_head=8, stack_num=2, ngram=4, seed=77)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp = synth.make_inputs(cfg, B=B, seed=5, pmi=pmi)
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
    model.set_precision("bf16" if dtype == "bf16" else "fp32")
    gf = GraphedForward(model, harness.call_args(inp, dev))
    corpus = (texts * (B * n_batches // len(texts) + 1))[:B * n_batches]
    cache = TokenCache(vocab, corpus)
    pipe = PipelinedForward(gf, vocab, cfg.T, B, dev)
    str_batches = lambda: (corpus[i * B:(i + 1) * B] for i in range(n_batches))
    id_batches = lambda: (cache.batch(i * B, (i + 1) * B) for i in range(n_batches))
    res = {"batches": n_batches, "batch": B, "texts": len(texts), "mean_tokens": round(float(np.mean([len(r) for r in cache.rows])), 1)}
    sums = {}
    for name, fn, src, ids in (("serial_strings", pipe.run_serial, str_batches, False), ("pipelined_strings", pipe.run, str_batches, False),
                               ("serial_cached_ids", pipe.run_serial, id_batches, True), ("pipelined_cached_ids", pipe.run, id_batches, True)):
        acc = torch.zeros(cfg.NL, device=dev, dtype=torch.float64)
        hook = lambda i, out: acc.add_(out.double().sum(0))
        fn(src(), from_ids=ids, on_logits=hook)                      # warm
        torch.cuda.synchronize()
        acc.zero_()
        t0 = time.perf_counter()
        n = fn(src(), from_ids=ids, on_logits=hook)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"samples_per_s": round(n * B / dt, 1), "ms_per_batch": round(dt / n * 1e3, 4)}
        sums[name] = acc.cpu().tolist()
    # two forwards in flight (graph.GraphedPipeline): the host pipeline feeding captures with buffers of their own
    gp = GraphedPipeline.of([gf, GraphedForward(model, harness.call_args(inp, dev), mode="segments")]) if gf.mode == "segments" else None
    if gp is not None:
        for name, src, ids in (("in_flight2_cached_ids", id_batches, True),):
            acc = torch.zeros(cfg.NL, device=dev, dtype=torch.float64)
            hook = lambda i, out: acc.add_(out.double().sum(0))
            pipe.run_in_flight(gp, src(), from_ids=ids, on_logits=hook)
            torch.cuda.synchronize()
            acc.zero_()
            t0 = time.perf_counter()
            n = pipe.run_in_flight(gp, src(), from_ids=ids, on_logits=hook)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[name] = {"samples_per_s": round(n * B / dt, 1), "ms_per_batch": round(dt / n * 1e3, 4)}
            sums[name] = acc.cpu().tolist()
        for _ in range(6):
            gp.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_batches):
            gp.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["device_only_in_flight2"] = {"samples_per_s": round(n_batches * B / dt, 1), "ms_per_batch": round(dt / n_batches * 1e3, 4)}
    # device-only rate of the same forward (inputs resident): the ceiling
    for _ in range(5):
        gf.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_batches):
        gf.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["device_only"] = {"samples_per_s": round(n_batches * B / dt, 1), "ms_per_batch": round(dt / n_batches * 1e3, 4)}
    ref = sums["serial_strings"]
    res["logit_checksums_equal"] = all(np.allclose(v, ref, rtol=0, atol=1e-6 * max(1.0, max(abs(x) for x in ref))) for v in sums.values())
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=60)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    print(json.dumps(measure(n_batches=a.batches, B=a.batch)))
