"""Child rank of tests/test_sharded_gpu.py (started as a fresh process: it touches the GPU only after it has been spawned).

    python tests/sharded_worker.py forward <out.pt> <config> <B> <seed> <precision>
        RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; every rank on cuda:0 (one-GPU box); gloo rendezvous.
        Builds the product model (seeded weights), runs it EAGERLY on its contiguous slice of the seeded batch through
        mgnns_amd.sharded.ShardedForward (the product's sharding + the logits all-gather) and saves the gathered logits.
    python tests/sharded_worker.py forward_graph <out.pt> <config> <B> <seed> <precision>
        The same slice captured as hipGraphs (mgnns_amd.graph.GraphedForward) and replayed eight times, the DEVICE logits of every
        replay gathered over gloo right behind it (no host synchronisation in the loop: the shape of bench.py's timed region, the
        pattern that degraded to 20-230 ms per step before ShardedForward joined the device in front of a gloo collective);
        saves the last gathered logits, the mean and the slowest step.
    python tests/sharded_worker.py stress <out.pt> <n> <batch>
        configs[4]: this rank's blocks of mgnns_amd.stress.StressWorkload(rank, world) (no data-path collective: plan_shards).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    dev = "cuda:0"
    if mode == "stress":
        from mgnns_amd import stress
        n, batch = int(sys.argv[3]), int(sys.argv[4])
        res = stress.StressWorkload(rank, world, n=n, batch=batch, dev=dev).forward()
        torch.cuda.synchronize()
        torch.save({k: v.cpu() for k, v in res.items()}, out_path)
        return 0
    import torch.distributed as dist
    from mgnns_amd import harness, synth
    from mgnns_amd.sharded import ShardedForward, shard_bounds
    cfg_name, B, seed, precision = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.CONFIGS[cfg_name]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp = synth.make_inputs(cfg, B=B, seed=seed, pmi=pmi)          # the SAME global batch on every rank; each keeps its slice
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
    model.set_precision(precision)
    lo, hi = shard_bounds(B, world, rank)
    mine = {k: (v[lo:hi] if k != "label_query" else v) for k, v in inp.items()}
    # RCCL refuses two ranks on one device, so on this one-GPU box the collective is gloo's, on the host copy of the local logits
    # (ShardedForward gathers on whatever device the logits live); with one GPU per rank the same class gathers over RCCL
    extra = {}
    if mode == "forward_graph":
        import time
        from mgnns_amd.graph import GraphedForward
        with torch.no_grad():
            gf = GraphedForward(model, harness.call_args(mine, dev))
            fwd = ShardedForward(lambda *a: gf.replay())
            for _ in range(2):
                fwd()                                              # (gloo's staging buffers, first-use costs: untimed)
            dist.barrier()
            worst, steps = 0.0, 8
            t_all = time.perf_counter()
            for _ in range(steps):
                t0 = time.perf_counter()
                gathered = fwd()
                worst = max(worst, time.perf_counter() - t0)
            mean = (time.perf_counter() - t_all) / steps
        extra = {"worst_step_ms": worst * 1e3, "mean_step_ms": mean * 1e3, "graph_mode": gf.mode}
    else:
        fwd = ShardedForward(lambda *a: model(*a).cpu())
        with torch.no_grad():
            gathered = fwd(*harness.call_args(mine, dev))
    torch.cuda.synchronize()
    torch.save(dict({"rank": rank, "lo": lo, "hi": hi, "logits": gathered.cpu()}, **extra), out_path)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
