#!/usr/bin/env python
"""The one-GPU two-rank hook of bench.py (MGNNS_BENCH_BACKEND=gloo MGNNS_BENCH_SAME_GPU=1) measures its SECOND leg at 11-220 ms
per step whichever leg that is.  Nothing of this repository's library runs here: two processes on one GPU,
a step = [a hipGraph of trivial kernels | the same kernels eagerly] + gloo's all_gather_into_tensor on DEVICE tensors, first
with one tensor size, then with another -- the shape of the hook's two legs.

    python tools/dev/two_proc_gloo.py [steps per phase] [graph|eager]
"""
import multiprocessing as mp
import os
import sys
import time


def child(rank, world, steps, mode, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    x = torch.zeros(65536, device=dev)
    res = []
    for phase, rows in enumerate((256, 128, 256)):
        local = torch.full((rows, 3), float(rank), device=dev)
        out = torch.empty(world * rows, 3, device=dev)

        def body():
            for _ in range(12):
                x.add_(1.0)

        run = body
        if mode == "graph":                            # a NEW capture per phase, like a new GraphedForward per leg
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                body()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                body()
            run = g.replay
        for _ in range(3):
            run()
            dist.all_gather_into_tensor(out, local)
        torch.cuda.synchronize()
        dist.barrier()
        t_run = t_gather = 0.0
        for _ in range(steps):
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dist.all_gather_into_tensor(out, local)
            torch.cuda.synchronize()
            t_run += t1 - t0
            t_gather += time.perf_counter() - t1
        assert float(out[-1, 0]) == world - 1
        # the same steps WITHOUT a host synchronisation between the launches and the collective (bench.py's timed region)
        loops = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                run()
                dist.all_gather_into_tensor(out, local)
            dist.barrier()
            torch.cuda.synchronize()
            loops.append(1e3 * (time.perf_counter() - t0) / steps)
        res.append((phase, rows, 1e3 * t_run / steps, 1e3 * t_gather / steps, loops))
    q.put((rank, res))
    dist.destroy_process_group()


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=child, args=(r, 2, steps, mode, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    for rank, res in sorted(q.get(timeout=600) for _ in ps):
        for phase, rows, a, b, loops in res:
            print("%s, rank %d, phase %d (%d rows per rank): launches + sync %.3f ms, gloo all-gather of device tensors %.3f ms per step; "
                  "five regions without the sync in between: %s ms per step"
                  % (mode, rank, phase, rows, a, b, " ".join("%.2f" % x for x in loops)), flush=True)
    for p in ps:
        p.join(60)
