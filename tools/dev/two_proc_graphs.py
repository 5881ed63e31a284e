#!/usr/bin/env python
"""Do hipGraph replays of two PROCESSES on one GPU degrade whatever the graph holds?  (the two-rank hook on the
one-GPU box measured 11-220 ms per step with graph replays against 2.3-2.5 ms with eager launches.)

    python tools/dev/two_proc_graphs.py [replays] [kernels per graph]

A graph of N trivial elementwise kernels on a 256-KB tensor (torch.cuda.CUDAGraph: nothing of this repository's library is
involved), replayed with a host synchronisation per replay like a training / serving step; the same launches eagerly; one
process alone, then two at once (children are spawned BEFORE they touch the GPU and meet at a barrier).  Prints the mean us
per replay of the first / middle / last third of the run and the worst replay, per process.
"""
import multiprocessing as mp
import os
import sys
import time


def child(rank, world, replays, nk, barrier, q, use_graph):
    import torch
    dev = torch.device("cuda:0")
    x = torch.zeros(65536, device=dev)
    s = torch.cuda.Stream()

    def body():
        for _ in range(nk):
            x.add_(1.0)

    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        g = None
        if use_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                body()
        run = g.replay if use_graph else body
        for _ in range(50):
            run()
        torch.cuda.synchronize()
        barrier.wait()
        ts = []
        for _ in range(replays):
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    n = len(ts) // 3
    us = lambda a: 1e6 * sum(a) / max(1, len(a))
    q.put((rank, us(ts[:n]), us(ts[n:2 * n]), us(ts[2 * n:]), 1e6 * max(ts)))


def leg(world, replays, nk, use_graph):
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(world), ctx.Queue()
    ps = [ctx.Process(target=child, args=(r, world, replays, nk, barrier, q, use_graph)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(60)
    for r, a, b, c, worst in out:
        print("%s, %d process%s, rank %d: %.1f / %.1f / %.1f us per replay (first / middle / last third), worst %.0f us"
              % ("hipGraph replays" if use_graph else "eager launches ", world, "es" if world > 1 else "", r, a, b, c, worst), flush=True)


if __name__ == "__main__":
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    nk = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("%d replays of %d trivial kernels, host synchronisation per replay" % (replays, nk))
    for world in (1, 2):
        for use_graph in (True, False):
            leg(world, replays, nk, use_graph)
