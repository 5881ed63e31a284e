#!/usr/bin/env python
"""The product forward replayed by one and by two PROCESSES on one GPU (11-220 ms per step for the second leg
of the two-rank hook).  tools/dev/two_proc_graphs.py shows that trivial graphs of two processes do NOT degrade; this tool says
which of the forward's kernels do:

    [MGNNS_FUSED_LABEL_GCN=0] [MGNNS_FUSED_LABEL_TAIL=0] python tools/dev/two_proc_model.py [replays] [batch] [graph|eager]

The persistent label GCN and the fused label tail are the launches whose workgroups WAIT for each other (work items of a layer
for the layer in front, ranks of a cluster for their partners' slices): they assume that every workgroup of the launch is
resident, which one process per GPU guarantees (grids <= 96 workgroups) and two processes on one GPU do not.
Children are spawned before they touch the GPU, meet at a barrier, then replay with a host synchronisation per forward.
"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(rank, world, replays, B, mode, barrier, q):
    sys.path.insert(0, ROOT)
    import torch
    from mgnns_amd import harness, synth
    from mgnns_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    cfg = synth.CONFIGS["mvsa_multiple_b256"]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp = synth.make_inputs(cfg, B=B, seed=cfg.seed + rank, pmi=pmi)
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
    model.set_precision("bf16").set_attention("faithful")
    call = harness.call_args(inp, dev)
    with torch.no_grad():
        if mode == "graph":
            gf = GraphedForward(model, call)
            run = gf.replay
        else:
            run = lambda: model(*call)
        for _ in range(10):
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
    ms = lambda a: 1e3 * sum(a) / max(1, len(a))
    q.put((rank, ms(ts[:n]), ms(ts[n:2 * n]), ms(ts[2 * n:]), 1e3 * max(ts)))


def leg(world, replays, B, mode):
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(world), ctx.Queue()
    ps = [ctx.Process(target=child, args=(r, world, replays, B, mode, barrier, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=900) for _ in ps)
    for p in ps:
        p.join(60)
    for r, a, b, c, worst in out:
        print("%s, B=%d, %d process%s, rank %d: %.3f / %.3f / %.3f ms per forward (first / middle / last third), worst %.2f ms"
              % (mode, B, world, "es" if world > 1 else "", r, a, b, c, worst), flush=True)


if __name__ == "__main__":
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    mode = sys.argv[3] if len(sys.argv) > 3 else "graph"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("fused label GCN %s, fused label tail %s" % (os.environ.get("MGNNS_FUSED_LABEL_GCN", "1"), os.environ.get("MGNNS_FUSED_LABEL_TAIL", "1")))
    for world in (1, 2):
        leg(world, replays, B, mode)
