#!/usr/bin/env python
"""bench.py's one-GPU two-rank hook, leg by leg, with the step split into graph replay and gloo gather (the
second leg of that hook reads 11-220 ms per step).  Two processes on one GPU; per leg a NEW GraphedForward of the product model
at the leg's batch + ShardedForward over gloo, like bench.py's measure():

    python tools/dev/two_proc_legs.py [steps] [batches, e.g. 256,128,256] [keep|drop] [mode: auto|segments|single]

keep / drop: whether the previous leg's capture stays alive.
"""
import gc
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(rank, world, steps, batches, keep, mode, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mgnns_amd import harness, synth
    from mgnns_amd.graph import GraphedForward
    from mgnns_amd.sharded import ShardedForward
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    cfg = synth.CONFIGS["mvsa_multiple_b256"]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp0 = synth.make_inputs(cfg, B=256, seed=cfg.seed + rank, pmi=pmi)
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp0["label_query"], dev)
    model.set_precision("bf16").set_attention("faithful")
    res, alive = [], []
    with torch.no_grad():
        for leg, B in enumerate(batches):
            sub = {k: (v[:B] if k != "label_query" else v) for k, v in inp0.items()}
            call = harness.call_args(sub, dev)
            gf = GraphedForward(model, call, mode=None if mode == "auto" else mode)
            sf = ShardedForward(lambda *a: gf.replay())
            for _ in range(3):
                sf.gather(gf.replay())
            torch.cuda.synchronize()
            dist.barrier()
            t_run = t_gather = 0.0
            worst = 0.0
            for _ in range(steps):
                t0 = time.perf_counter()
                o = gf.replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sf.gather(o)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                t_run += t1 - t0
                t_gather += t2 - t1
                worst = max(worst, t2 - t0)
            # the same steps WITHOUT a host synchronisation in between (bench.py's timed region: barrier, K steps, barrier)
            loops = []
            for _ in range(5):
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    sf.gather(gf.replay())
                dist.barrier()
                torch.cuda.synchronize()
                loops.append(1e3 * (time.perf_counter() - t0) / steps)
            res.append((leg, B, gf.mode, 1e3 * t_run / steps, 1e3 * t_gather / steps, 1e3 * worst, loops))
            if keep:
                alive.append((gf, sf))
            else:
                del gf, sf
                gc.collect()
    q.put((rank, res))
    dist.destroy_process_group()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "256,128,256").split(",")]
    keep = (sys.argv[3] if len(sys.argv) > 3 else "keep") == "keep"
    mode = sys.argv[4] if len(sys.argv) > 4 else "auto"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=child, args=(r, 2, steps, batches, keep, mode, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    for rank, res in sorted(q.get(timeout=900) for _ in ps):
        for leg, B, m, a, b, w, loops in res:
            print("rank %d leg %d B=%d [%s, previous captures %s]: replay + sync %.3f ms, gloo gather %.3f ms per step, worst step %.2f ms; "
                  "five regions without a sync per step: %s ms per step"
                  % (rank, leg, B, m, "kept" if keep else "dropped", a, b, w, " ".join("%.2f" % x for x in loops)), flush=True)
    for p in ps:
        p.join(60)
