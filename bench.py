#!/usr/bin/env python
"""bench.py -- forward samples/s of the MGNNS hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256] [--dtype f32] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward of the whole model (text GCN + BiLSTM text bank + object/scene GCN channels +
label attention + 4 stacks of single-query multi-head fusion + classifier) over one synthetic
MVSA-Multiple-shaped batch (configs[2]: B=256, T=100, V=20154, 8 heads, 2 layers), entered at
the [B,2048,14,14] feature maps, inputs resident in HBM.

N>1: one process per GPU.  `python bench.py --gpus N` WITHOUT a torchrun environment starts the N ranks itself
(child processes, the parent never touches a GPU) and fails non-zero unless all N join; under torchrun it is one
rank.  The batch shards over ranks with replicated weights and ONE collective, the RCCL all-gather of the logits,
inside the timed region (a node of the step's hipGraph when the capture probe passes, else right behind the
replay).  Both scalings are timed: weak (256 samples per GPU, the headline `value`, "scaling": "weak") and strong
(configs[3]: the global B=256 batch split 256/N per GPU, reported under "strong_scaling"); `--scaling strong` makes
the strong figure the headline.  Rank 0 prints ONE JSON line.

`--dry-launch` runs the same launcher / rank seeding / barrier + MAX-over-ranks timing / single-line protocol on
the gloo backend with no GPU and no model (tests/test_bench_launch_cpu.py).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0}       # MI355X_MICROARCH.md: dense MFMA peaks
PEAK_HBM_GBPS = 8000.0
GLOBAL_BATCH = 256                                  # BASELINE.json: "forward samples/sec at batch 256"
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of --steps steps each (one warm-up, then the regions back to back, each bracketed by "
                         "barrier + synchronize): ms_per_step / value are the MEDIAN region, min / max are reported next to it")
    ap.add_argument("--batch", type=int, default=GLOBAL_BATCH, help="samples per GPU (weak scaling)")
    ap.add_argument("--config", default="mvsa_multiple_b256")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="which figure is the headline `value` at N>1 (both are always measured)")
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "bf16x3"],
                    help="bf16 = BASELINE configs[2] (bf16 MFMA operands, fp32 accumulate); f32 = exact-f32 MFMA parity path")
    ap.add_argument("--attn", default="faithful", choices=["auto", "faithful", "folded"],
                    help="fusion attention: faithful = K/V projected from the memory bank as the reference does (the headline: the "
                         "formulation the north-star's MFMA figure is quoted on); folded = the projections folded away algebraically "
                         "(one read of the memory bank; always reported next to the headline as a variant and in roofline_all); "
                         "auto = folded in bf16 mode, faithful in fp32 mode")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra legs (fp32 parity mode, folded attention, "
                                                               "configs[4] stress, CNN trunks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph replay per step")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="forwards in flight: captures of the forward replayed round robin without a join between them "
                         "(mgnns_amd.graph.GraphedPipeline); 1 = one capture, every replay joins the four streams before the next")
    ap.add_argument("--single-stream", action="store_true",
                    help="with --no-graph: every kernel on one stream (no concurrent kernels) -- the setting the rocprofv3 "
                         "per-kernel averages under profiles/ are taken in, comparable with roofline.avg_launch_ms")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher / timing protocol only: gloo backend, no GPU, no model (CPU test of the N-rank path)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks are stopped after this many seconds")
    ap.add_argument("--probe-collective", action="store_true", help=argparse.SUPPRESS)     # internal: capture probe child
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without WORLD_SIZE starts the ranks itself
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Parent of a self-launched run.  Starts one child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its
    environment, exactly what torch.distributed.run would set), passes rank 0's stdout through, and returns non-zero
    unless every rank exits 0 and the JSON line reports n_gpus == N.  This process makes no GPU call."""
    n = args.gpus
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "MGNNS_BENCH_SELF_LAUNCHED": "1"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + args.launch_timeout
    failed = []
    try:
        while any(p.poll() is None for p in procs):
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad:                         # a dead rank leaves the others in a barrier: stop them now
                failed += bad
                break
            if time.time() > deadline:
                failed.append(("timeout_s", args.launch_timeout))
                break
            time.sleep(0.2)
    finally:
        for p in procs:                     # exactly the PIDs started here
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    failed += [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0 and (r, p.returncode) not in failed]
    reader.join(timeout=10)
    out0 = "".join(c for c in chunks if c)
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{"):
            try:
                line = json.loads(ln)
            except ValueError:
                pass
        else:
            print(ln, file=sys.stderr)
    if failed:
        print("bench.py: %d-rank launch failed: %s" % (n, failed), file=sys.stderr)
        return 1
    if line is None or line.get("n_gpus") != n:
        print("bench.py: rank 0 reported n_gpus=%r, expected %d" % (None if line is None else line.get("n_gpus"), n),
              file=sys.stderr)
        return 1
    print(json.dumps(line), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------------------
# protocol shared by the real and the dry run
# ------------------------------------------------------------------------------------------------------------------
def timed_steps(step, steps, warmup, barrier, after_warmup=None):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier() on both sides -> local seconds."""
    for _ in range(warmup):
        step()
    if after_warmup is not None:
        after_warmup()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(dt, dist, device):
    import torch
    if dist is None:
        return dt, [dt]
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), [float(e.item()) for e in every]


def rank_seed(base, rank, scaling):
    """Weak scaling: every rank draws its own shard (seed + 1000*rank).  Strong: all ranks draw the SAME global batch
    (seed) and keep their contiguous slice, so the gathered logits equal the single-GPU logits of that batch."""
    return base + 1000 * rank if scaling == "weak" else base


def flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


# ------------------------------------------------------------------------------------------------------------------
# the single JSON line: compact (a few KB) on the LAST stdout line, every detail in bench_detail.json
# ------------------------------------------------------------------------------------------------------------------
LINE_TARGET_BYTES = 4000                 # what the line is kept under
LINE_LIMIT_BYTES = 8000                  # hard limit (asserted): round 5's 21.7 KB line was not parsed by the driver
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")
STR_LIMIT = 150                          # strings of the compact line (the full text stays in the detail file)
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "global_batch", "parallelism", "launch", "attention", "forwards_in_flight", "schedule",
               "collective", "sources")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed",
                 "avg_launch_ms_rocprof", "frac_rocprof", "avg_launch_ms_back_to_back", "flops_per_launch",
                 "bytes_per_launch")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "cpu_model", "host_logical_cpus", "single_thread_samples_per_s")
SCALING_KEYS = ("value", "unit", "ms_per_step", "per_gpu_batch", "global_batch")


def _short(v):
    return v if not isinstance(v, str) or len(v) <= STR_LIMIT else v[:STR_LIMIT - 3] + "..."


def _strict(v):
    """NaN / +-inf -> null, recursively: the line must parse under a strict JSON reader."""
    if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
        return None
    if isinstance(v, dict):
        return {k: _strict(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_strict(x) for x in v]
    return v


def _pick(d, keys):
    return None if d is None else {k: _short(d[k]) for k in keys if k in d}


def compact_line(line):
    """The full result object -> (compact, detail).  `compact` is what the driver parses: the contract's keys, `roofline`,
    `cpu_baseline`, the parity figure and the scalar value_* / max_abs_logit_diff_* / *_one_in_flight keys; everything else
    (roofline_all, variants, small_batch, stress, text_pipeline, trunks, timing, prose) lives in `detail` = the full object."""
    c = {k: line[k] for k in HEAD_KEYS if k in line}
    c["config"] = _pick(line.get("config") or {}, CONFIG_KEYS)
    c["roofline"] = _pick(line.get("roofline"), ROOFLINE_KEYS)
    c["cpu_baseline"] = _pick(line.get("cpu_baseline"), CPU_KEYS)
    for k, v in line.items():
        if k in c:
            continue
        scalar = v is None or isinstance(v, (bool, int, float))
        if scalar and (k.startswith(("value_", "max_abs_logit_diff", "ms_per_step_")) or k == "dry_launch"):
            c[k] = v
    for k in ("weak_scaling", "strong_scaling"):
        if isinstance(line.get(k), dict):
            c[k] = _pick(line[k], SCALING_KEYS)
    c["detail"] = os.path.basename(DETAIL_FILE)
    return c, line


def summary_of(line):
    """A numbers-only digest of the detail legs (no prose), printed as the `bench_summary` line right above the result line:
    what a reader of the last few KB of stdout sees without opening bench_detail.json."""
    out = {}
    ra = line.get("roofline_all")
    if ra:          # [kernel (short), bound, avg us, frac]
        out["roofline_all"] = [[r.get("kernel", "")[:40], r.get("bound"), r.get("avg_us"), r.get("frac")] for r in ra]
    v = line.get("variants")
    if v:           # name (short) -> [samples/s, ms, in flight, |dlogit|]
        out["variants"] = {k[:44]: [r.get("value"), r.get("ms_per_step"), r.get("forwards_in_flight"),
                                    r.get("max_abs_logit_diff_vs_cpu_oracle")] for k, r in v.items()}
    sb = line.get("small_batch")
    if sb:          # B -> [ms one at a time, ms two in flight]
        out["small_batch_ms"] = {k: [r.get("ms_per_step"), r.get("ms_per_step_in_flight2")] for k, r in sb.items()
                                 if isinstance(r, dict) and "ms_per_step" in r}
    st = line.get("stress")
    if isinstance(st, dict):
        d = {}
        for k, r in st.items():
            if not isinstance(r, dict):
                continue
            if "cold_ms" in r:
                d[k] = [r["cold_ms"], r.get("frac_of_8TBps", r.get("frac_of_bf16_mfma_peak"))]
            elif "ms_per_3_channel_forward" in r:
                d[k] = [r["ms_per_3_channel_forward"], r.get("ms_gcn_of_one_channel"),
                        r.get("ms_per_3_channel_forward_graphs_on_3_streams")]
                if "max_abs_diff_vs_reference_order_over_output_scale" in r:
                    d[k].append(r["max_abs_diff_vs_reference_order_over_output_scale"])
        out["stress"] = d if d else st
    tp = line.get("text_pipeline")
    if isinstance(tp, dict):
        out["text_pipeline_ms"] = {k: r["ms_per_batch"] for k, r in tp.items() if isinstance(r, dict) and "ms_per_batch" in r} or tp
    tr = line.get("trunks")
    if isinstance(tr, dict):
        out["trunks"] = {k: [r["ms_per_batch"], r.get("frac_of_bf16_mfma_peak")] for k, r in tr.items()
                         if isinstance(r, dict) and "ms_per_batch" in r} or tr
    if "timing" in line:
        out["regions_ms_per_step"] = line["timing"].get("regions_ms_per_step")
    return out


def emit(line):
    """Write the full object to bench_detail.json (and gpurun_out/ when that exists), echo it on an EARLIER stdout line that
    does not start with '{', then print the compact object as the last stdout line."""
    line = _strict(line)
    blob = json.dumps(line, allow_nan=False)
    flush_c_stdio()          # whatever native libraries (RCCL banner) still hold in the C stdio buffer goes out first
    sys.stdout.flush()
    if len(blob) <= LINE_TARGET_BYTES and line.get("dry_launch"):       # protocol-only runs: already compact, nothing to move
        print(blob, flush=True)
        return
    compact, detail = compact_line(line)
    text = json.dumps(compact, allow_nan=False)
    assert len(text) < LINE_LIMIT_BYTES, "result line is %d bytes (limit %d)" % (len(text), LINE_LIMIT_BYTES)
    for path in (DETAIL_FILE, os.path.join(ROOT, "gpurun_out", os.path.basename(DETAIL_FILE))):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(blob + "\n")
        except OSError as e:
            print("bench.py: could not write %s (%s)" % (path, e), file=sys.stderr)
    print("bench_detail " + blob, flush=True)
    digest = json.dumps(summary_of(line))
    if len(digest) > 2 and len(digest) + len(text) < LINE_LIMIT_BYTES:
        print("bench_summary " + digest, flush=True)
    print(text, flush=True)


def dry_run(args, rank, world):
    """The N-rank protocol without a GPU: gloo, a step that sleeps (rank+1) ms and 'logits' that encode rank and seed."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from mgnns_amd.sharded import ShardedForward, shard_bounds
    d = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        d = dist
    res = {}
    for scaling in ("weak", "strong"):
        if scaling == "weak":
            b_local, seed = args.batch, rank_seed(1237, rank, "weak")
            local = np.random.RandomState(seed).standard_normal((b_local, 3)).astype(np.float32)
        else:
            lo, hi = shard_bounds(GLOBAL_BATCH, world, rank)
            b_local, seed = hi - lo, rank_seed(1237, rank, "strong")
            local = np.random.RandomState(seed).standard_normal((GLOBAL_BATCH, 3)).astype(np.float32)[lo:hi]
        sf = ShardedForward(lambda x: x)
        x = torch.from_numpy(local)
        state = {}

        def step():
            time.sleep(1e-3 * (rank + 1))
            state["out"] = sf(x)

        def barrier():
            if d is not None:
                d.barrier()

        dt = timed_steps(step, args.steps, args.warmup, barrier)
        dt_max, dts = max_over_ranks(dt, d, "cpu")
        seeds = [seed]
        if d is not None:
            seeds = [None] * world
            d.all_gather_object(seeds, seed)
        total = world * b_local if scaling == "weak" else GLOBAL_BATCH
        res[scaling] = {"value": total / (dt_max / args.steps), "ms_per_step": dt_max / args.steps * 1e3, "dt_ranks": dts,
                        "seeds": seeds, "gathered_rows": int(state["out"].shape[0]),
                        "gathered_checksum": float(state["out"].double().sum())}
    if d is not None:
        d.barrier()
        d.destroy_process_group()
    if rank != 0:
        return
    head = res[args.scaling]
    line = {"metric": "forward samples/sec at batch 256 (3-channel GCN + fusion)", "value": round(head["value"], 1),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "none", "data": "synthetic", "dry_launch": True,
            "config": {"workload": "dry launch (gloo, no GPU, no model): protocol check only"},
            "weak_scaling": res["weak"], "strong_scaling": res["strong"]}
    emit(line)


def stress_run(args, rank, local_rank, world):
    """`--config stress`: BASELINE configs[4] -- three 10 000-node label-graph channels, batch 512, bf16 -- over `world` ranks.
    The channels are independent and the read-out is per sample, so the work shards with NO data-path collective
    (mgnns_amd/stress.py::plan_shards: whole channels up to three ranks, the read-out batch of a channel beyond); a step is one
    forward of the rank's share, bracketed by barriers, MAX over ranks; value = read-out samples per second of the whole job
    (strong scaling: the 3 x 512 job is fixed).  --dry-launch: the same protocol on gloo with no GPU (the step sleeps)."""
    from mgnns_amd import stress
    dry = args.dry_launch
    dist = None
    if world > 1:
        import torch.distributed as dist
        import torch
        backend = "gloo" if dry else os.environ.get("MGNNS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    plan = stress.plan_shards(world)
    mine = plan[rank]
    n_nodes = int(os.environ.get("MGNNS_STRESS_NODES", stress.N_NODES))
    density = float(os.environ.get("MGNNS_STRESS_DENSITY", stress.DENSITIES[0]))
    if dry:
        device = "cpu"

        def step():
            time.sleep(1e-3 * sum((b1 - b0) / stress.BATCH for _, b0, b1 in mine))      # 1 ms per full channel

        def barrier():
            if dist is not None:
                dist.barrier()
        blocks = [(c, b0, b1, b1 - b0, n_nodes) for c, b0, b1 in mine]
    else:
        import torch
        if os.environ.get("MGNNS_BENCH_SAME_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        wl = stress.StressWorkload(rank, world, n=n_nodes, density=density, dev=device)
        state = {}

        def step():
            state["out"] = wl.forward()

        def barrier():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
        step()
        blocks = [(c, b0, b1) + tuple(v.shape) for (c, b0, b1), v in state["out"].items()]
    dt = timed_steps(step, args.steps, args.warmup, barrier)
    dt_max, dts = max_over_ranks(dt, dist, device)
    every = [blocks]
    if dist is not None:
        every = [None] * world
        dist.all_gather_object(every, blocks)
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    covered = sum(rows for rb in every for (_, _, _, rows, _) in rb)
    assert covered == stress.N_CHANNELS * stress.BATCH, "shards cover %d of %d channel-samples" % (covered, stress.N_CHANNELS * stress.BATCH)
    ms = dt_max / args.steps * 1e3
    line = {"metric": "configs[4] stress: read-out samples/sec of the 3-channel 10k-node label GCN (batch 512 per channel)",
            "value": round(stress.N_CHANNELS * stress.BATCH / ms * 1e3, 1), "unit": "channel-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "none" if dry else "bf16", "data": "synthetic", "dry_launch": bool(dry),
            "config": {"workload": "configs[4]: %d-node graph, density %g CSR, 3 channels, batch %d, bf16 / fp32 accumulate; "
                                   "sharded by channel, then read-out batch; no collective" % (n_nodes, density, stress.BATCH)},
            "dt_ranks": dts, "shards": [[list(b[:3]) for b in rb] for rb in every],
            "block_shapes": [[list(b[3:]) for b in rb] for rb in every]}
    emit(line)


# ------------------------------------------------------------------------------------------------------------------
# in-graph collective probe (child process, so a hang or a crash cannot take the benchmark down)
# ------------------------------------------------------------------------------------------------------------------
def probe_collective_child():
    """Capture an RCCL all-gather into a hipGraph, replay it on changing inputs, check the result; exit 0 iff it works."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    x = torch.full((32, 3), float(rank), device=dev)
    out = torch.empty(world * 32, 3, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            dist.all_gather_into_tensor(out, x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dist.all_gather_into_tensor(out, x)
    ok = True
    for it in range(3):
        x.fill_(float(rank + 10 * it))
        g.replay()
        torch.cuda.synchronize()
        want = torch.arange(world, device=dev, dtype=torch.float32).repeat_interleave(32)[:, None].expand(-1, 3) + 10 * it
        ok = ok and bool(torch.equal(out, want))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def probe_collective(timeout=240.0):
    """Run probe_collective_child with this rank's coordinates on a port of its own.  Called BEFORE this process touches
    the GPU.  True iff the child exits 0 in time."""
    env = dict(os.environ)
    for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1")):
        env.setdefault(k, v)                     # a single rank run with MGNNS_FORCE_DIST=1 has no torchrun environment
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + 17)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-collective"], env=env, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print("collective-capture probe failed (rc %d): %s" % (r.returncode, r.stderr[-400:]), file=sys.stderr)
        return r.returncode == 0
    except subprocess.TimeoutExpired:
        print("collective-capture probe timed out", file=sys.stderr)
        return False


# ------------------------------------------------------------------------------------------------------------------
# algorithmic work -- what every roofline figure below is computed from
# ------------------------------------------------------------------------------------------------------------------
def mha_core_flops(B, L, D, H, dk):
    """One fused single-query MHA launch: K and V projections 2*(2*L*D*H*dk) + QK^T and PV 2*(2*H*dk*L) per sample."""
    return B * (4.0 * L * D * H * dk + 4.0 * H * dk * L)


def textgcn_bytes(tok, ngram, D=300):
    """sum_b [U_b*4D + E_b*12 + 4D]: gathered node rows, (edge id, weight, index) per edge, the output row.
    U_b = distinct token ids of the document (PAD node included when padded), E_b = n-gram window edges + self loops."""
    import numpy as np
    total = 0.0
    for row in tok:
        n = int((row != 0).sum())
        U = len(np.unique(row))
        i = np.arange(n)
        E = int((np.minimum(i, ngram) + np.minimum(n - 1 - i, ngram) + 1).sum())
        total += U * 4.0 * D + E * 12.0 + 4.0 * D
    return total


def sha16(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (cannot be collected live); refused when the kernel
    source changed since the pass (sha of the .hip file stored with the entry)."""
    try:
        with open(PMC_TRAFFIC) as f:
            t = json.load(f).get(kernel_key)
        if not t:
            return None, None
        src = os.path.join(ROOT, t["kernel_source"])
        if sha16(src) != t["kernel_source_sha16"]:
            return None, "stale: %s changed since %s" % (t["kernel_source"], t["source"])
        return t["hbm_bytes"], t["source"]
    except (OSError, ValueError, KeyError):
        return None, None



def core_back_to_back(model, cfg, B, P, dtype, n=40, reps=5):
    """The dominant attention kernel timed as `n` launches captured back to back in one hipGraph between ONE pair of HIP events
    (no per-launch event / launch overhead in the figure: what rocprofv3's kernel durations average to): random bank and query
    of the forward's shapes, the model's own packed weights of the first text->object layer.  -> ms per launch, or None."""
    import torch
    from mgnns_amd import ops
    try:
        a = model.text_img_object_multi_head_att[0].slf_attn
        dev = a.w_ks.weight.device
        g = torch.Generator(device=dev).manual_seed(11)
        bank = torch.randn(B, P, cfg.emb_size, device=dev, generator=g)
        qh = torch.randn(B, cfg.n_head * cfg.d_kv, device=dev, generator=g)
        bk, bv = a.w_ks.bias.detach(), a.w_vs.bias.detach()
        if dtype == "bf16":
            bb, wp = ops.cast_pad_bf16(bank), a._packed_kv(ops.MHA_CORE_PLAIN)
            fn = lambda: ops.sq_mha_core_bf16(qh, bb, None, cfg.n_head, cfg.d_kv, wp, bk, bv, want_attn=False)
        elif dtype == "bf16x3":
            sp, wp = ops.split_pad_bf16(bank), a._packed_kv("split")
            fn = lambda: ops.sq_mha_core_split(qh, sp, None, cfg.n_head, cfg.d_kv, wp, bk, bv, want_attn=False)
        else:
            wk, wv = a.w_ks.weight.detach(), a.w_vs.weight.detach()
            fn = lambda: ops.sq_mha_core(qh, bank, None, cfg.n_head, cfg.d_kv, wk, bk, wv, bv, want_attn=False)
        st = torch.cuda.Stream(device=dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st), torch.no_grad():
            for _ in range(3):
                fn()
            st.synchronize()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(n):
                    fn()
            gr.replay()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                gr.replay()
            e1.record(st)
            st.synchronize()
        return e0.elapsed_time(e1) / (reps * n)
    except Exception as e:            # a measurement aid: never the reason a bench line is lost
        print("core_back_to_back: %s: %s" % (type(e).__name__, e), file=sys.stderr)
        return None


def rocprof_avg_us(kernel_substr, mode="bf16_serial"):
    """(avg us, file) of a kernel in the newest committed rocprofv3 summary of that mode under profiles/ (rNN_<mode>_kernel_stats.md);
    (None, None) when there is none.  Cross-check only: it is whatever box the profile ran on."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_kernel_stats.md" % mode)))
    for f in reversed(files):
        try:
            for ln in open(f):
                if ln.startswith("| `") and kernel_substr in ln:
                    cols = [c.strip() for c in ln.strip().strip("|").split("|")]
                    return float(cols[3]), os.path.relpath(f, ROOT)
        except (OSError, ValueError, IndexError):
            continue
    return None, None

def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, model, inp, pmi, budget_s=30.0):
    """The CPU oracle (oracle/restatement.py, the pinned restatement of the reference forward) timed on this box's host
    cores on the SAME synthetic batch.  Checker/baseline only -- never the product.  Thread sweep on a 64-sample slice,
    then 3 warm-up + >=5 timed forwards of the full batch at the best thread count, plus a 1-thread figure."""
    import numpy as np
    import torch
    from oracle import restatement as R
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ti = {k: torch.as_tensor(v) for k, v in inp.items()}
    lq = model.label_query.detach().cpu()
    B = ti["text"].shape[0]
    ncpu = os.cpu_count() or 1
    t_begin = time.time()

    def run(t, n):
        sub = {k: (v[:n] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in t.items()}
        return R.forward(p, sub, pmi, cfg.n_head, cfg.d_kv, cfg.stack_num, cfg.ngram, label_query=lq)

    def best_of(n, reps):
        ts = []
        for _ in range(reps):
            t0 = time.time()
            run(ti, n)
            ts.append(time.time() - t0)
        return min(ts)

    nsub = min(B, 64)
    sweep = {}
    # thread counts up to 64 (+ all cores only on hosts with <= 64: on the 256-thread GPU box torch's intra-op pool with
    # every logical CPU ran this workload at 0.3 samples/s -- ten minutes for one sweep point); the sweep stops early once a
    # point is 2x slower than the best so far
    cand = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | ({ncpu} if ncpu <= 64 else set()))
    for th in cand:
        torch.set_num_threads(th)
        run(ti, nsub)
        sweep[th] = round(nsub / best_of(nsub, 2), 1)
        if sweep[th] * 2 < max(sweep.values()):
            break
    best_th = max(sweep, key=sweep.get)
    torch.set_num_threads(1)
    n1 = min(B, 8)
    one = round(n1 / best_of(n1, 1), 2)
    torch.set_num_threads(best_th)
    ref = run(ti, B)                               # warm-up 1 (also the parity reference)
    per = time.time()
    run(ti, B)
    per = time.time() - per                        # warm-up 2 doubles as the cost estimate
    run(ti, B)
    left = budget_s - (time.time() - t_begin)
    reps = int(max(5, min(8, left / max(per, 1e-3))))
    times = []
    for _ in range(reps):
        t0 = time.time()
        run(ti, B)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return ref, {"value": round(B / med, 2), "unit": "samples/s", "cores": int(best_th), "kind": "port",
                 "cpu_model": cpu_model(), "host_logical_cpus": ncpu,
                 "sample": "3 warm-up + %d timed B=%d forwards of oracle/restatement.py (torch-CPU fp32, %d threads = best of "
                           "sweep; median %.3f s)" % (len(times), B, best_th, med),
                 "thread_sweep_samples_per_s": {str(k): v for k, v in sweep.items()},
                 "thread_sweep_sample": "B=%d slice of the same batch, best of 2 after 1 warm-up" % nsub,
                 "single_thread_samples_per_s": one}


def cpu_baseline_other(names=("mvsa_single_b8", "tumemo_b64")):
    """configs[0] / configs[1] shapes through the oracle (cfg 1 is the reference's own CPU-runnable case)."""
    import numpy as np
    import torch
    from mgnns_amd import harness, synth
    from oracle import restatement as R
    out = {}
    for name in names:
        cfg = synth.CONFIGS[name]
        pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
        A_obj, A_place = harness.synthetic_adjacencies(cfg)
        inp = synth.make_inputs(cfg, seed=cfg.seed, pmi=pmi)
        m = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], None)
        p = {k: v.detach() for k, v in m.state_dict().items()}
        ti = {k: torch.as_tensor(v) for k, v in inp.items()}
        f = lambda: R.forward(p, ti, pmi, cfg.n_head, cfg.d_kv, cfg.stack_num, cfg.ngram,
                              label_query=torch.as_tensor(inp["label_query"]))
        for _ in range(2):
            f()
        ts = []
        for _ in range(5):
            t0 = time.time()
            f()
            ts.append(time.time() - t0)
        out[name] = {"samples_per_s": round(cfg.B / float(np.median(ts)), 1), "batch": cfg.B, "timed": 5,
                     "threads": torch.get_num_threads()}
    return out


# ------------------------------------------------------------------------------------------------------------------
# extra legs (single GPU only, never part of `value`)
# ------------------------------------------------------------------------------------------------------------------
def trunk_leg(dev, batch=128, size=448, iters=3):
    """ResNet-101 (objects) + ResNet-50/365 (places) features of `batch` 448x448 images on the HIP
    implicit-GEMM kernels (seeded weights), eager launches; MFMA utilisation on the algorithmic convolution FLOPs."""
    import torch
    from mgnns_amd import synth
    try:
        from mgnns_amd import trunk
        res = {}
        for name, ctor in (("resnet101", trunk.resnet101), ("resnet50_places365", lambda: trunk.resnet50(365))):
            feats = trunk.ResNetFeatures(synth.fill_trunk_(ctor(), 3).eval()).to(dev).eval()
            img = torch.randn(batch, 3, size, size, device=dev)
            for _ in range(2):
                feats(img)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                feats(img)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            fl = trunk.features_flops(feats, size) * batch
            res[name] = {"ms_per_batch": round(ms, 3), "images_per_s": round(batch / ms * 1e3, 1),
                         "achieved_tflops": round(fl / ms / 1e9, 1), "frac_of_bf16_mfma_peak": round(fl / ms / 1e9 / 2500.0, 4)}
            del feats, img
        both = sum(v["ms_per_batch"] for v in res.values())
        res["what"] = ("the two CNN trunks in front of the path (MODEL:274-294), batch %d of %dx%d images each, bf16 NHWC "
                       "implicit-GEMM convolutions; not part of `value`" % (batch, size, size))
        res["images_per_s_both_trunks"] = round(batch / both * 1e3, 1)
        return res
    except Exception as e:     # the headline line must survive a failure of this extra leg
        return {"error": "%s: %s" % (type(e).__name__, e)}


def stress_leg(dev):
    """BASELINE configs[4] on this GPU: one 10 000-node channel, cache-cold (mgnns_amd/stress.py)."""
    try:
        from mgnns_amd import stress
        return stress.measure(dev)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def text_pipeline_leg(dev):
    """real texts -> ids -> pinned buffers -> H2D -> captured forward, serial vs two-deep pipelined
    (tools/bench_text_pipeline.py); reported beside the headline, never part of `value`."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_text_pipeline
        r = bench_text_pipeline.measure(str(dev), n_batches=40)
        r["what"] = ("end-to-end samples/s with the text side coming from the HOST per batch (302 real val-split texts, cycled): "
                     "word2id + padding into pinned buffers + async H2D + hipGraph replay; feature maps resident")
        return r
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def graphed_variant(model, call, B, steps, warmup, what, in_flight=1, regions=3):
    """One more captured form of the step, timed like the headline: `regions` regions of `steps` replays, the median reported."""
    import torch
    from mgnns_amd import _lib
    from mgnns_amd.graph import GraphedForward, GraphedPipeline

    def timed(fn, end):
        for _ in range(warmup):
            fn()
        ds = []
        for _ in range(max(1, regions)):
            end()
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(steps):
                fn()
            end()
            torch.cuda.synchronize()
            ds.append((time.perf_counter() - tv) / steps)
        _lib.take_status()
        return sorted(ds)[(len(ds) - 1) // 2], ds

    gv = GraphedForward(model, call, mode="segments" if in_flight > 1 else None)
    dv, dvs = timed(gv.replay, lambda: None)
    vout = gv.static_out
    r = {"value": round(B / dv, 1), "unit": "samples/s", "ms_per_step": round(dv * 1e3, 4), "forwards_in_flight": 1, "what": what,
         "regions_ms_per_step": [round(d * 1e3, 4) for d in dvs], "_out": vout[:B].float().cpu()}
    if in_flight > 1:       # the same measurement as the headline: captures with buffers of their own, replayed without a join
        pipe = GraphedPipeline.of([gv] + [GraphedForward(model, call, mode="segments") for _ in range(in_flight - 1)])
        dp, dps = timed(pipe.replay, pipe.wait)
        r["serial_replay"] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "regions_ms_per_step": r["regions_ms_per_step"]}
        r.update({"value": round(B / dp, 1), "ms_per_step": round(dp * 1e3, 4), "forwards_in_flight": in_flight,
                  "regions_ms_per_step": [round(d * 1e3, 4) for d in dps]})
    return r


def roofline_all(timer, cfg, B, P, inp, dtype, model):
    """Per-kernel achieved fraction of the bounding roofline from the same eager single-stream timing leg as `roofline`
    (HIP events around each C-ABI call on its launch stream).  Algorithmic bytes / FLOPs per SURVEY 8d."""
    import numpy as np
    d = timer.durations_ms()
    H, dk, D, T = cfg.n_head, cfg.d_kv, cfg.emb_size, cfg.T
    rows = []

    L2_PER_CU_GBPS = 64 * 2.4          # a CU fetches 64 B/clk from its XCD's L2 (MI355X_MICROARCH.md), 2.4 GHz

    def add(kernel, key, bound, work, note, pmc_key=None, l2_stream_bytes=None):
        v = d.get(key, [])
        if not v:
            return
        us = float(np.mean(v)) * 1e3
        if bound == "hbm":
            ach, peak, unit = work / us / 1e3, PEAK_HBM_GBPS, "GB/s"
        else:
            ach, peak, unit = work / us / 1e6, PEAK_TFLOPS["f32" if bound == "mfma_f32" else "bf16"], "TFLOP/s"
        row = {"kernel": kernel, "bound": "mfma" if bound.startswith("mfma") else bound, "algorithmic": work,
               "algorithmic_unit": "B" if bound == "hbm" else "FLOP", "avg_us": round(us, 2), "launches": len(v),
               "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "note": note}
        if pmc_key:                      # HBM bytes per launch of the committed PMC passes (refused when the kernel changed since)
            row["traffic"], row["traffic_source"] = pmc_traffic(pmc_key)
        if l2_stream_bytes:              # kernels whose workgroups stream packed weights from L2: the bound that matters for them
            row["l2_stream"] = {"bytes_per_workgroup_on_its_critical_path": int(l2_stream_bytes),
                                "per_cu_l2_rate_GBps": L2_PER_CU_GBPS,
                                "frac_of_per_cu_l2_rate": round(l2_stream_bytes / (us * 1e-6) / 1e9 / L2_PER_CU_GBPS, 4)}
        rows.append(row)

    pk = lambda n, k: ((n + 15) // 16) * ((k + 31) // 32) * 1024          # bytes of one packed bf16 image of an [n, k] weight
    bf = dtype == "bf16"
    core = "mgnns_sq_mha_core_bf16_fwd" if bf else "mgnns_sq_mha_core_fwd"
    mfma = "mfma_bf16" if bf else "mfma_f32"
    add("sq_mha_core%s (L=%d image bank)" % ("_bf16" if bf else "", P), (core, P, False), mfma,
        mha_core_flops(B, P, D, H, dk), "K/V projection + QK^T + softmax + PV, all rows valid")
    from mgnns_amd import ops as _ops
    if bf and _ops.MHA_CORE == 32 and _ops.MHA_PACKED and T <= _ops.PLAN_MAX_L:
        live = float(np.asarray(inp["text_mask"]).sum())
        add("sq_mha32_packed (L=%d text bank, masked: the live rows of short samples share a workgroup)" % T, (core, T, True), mfma,
            mha_core_flops(B, T, D, H, dk) * live / (B * T),
            "FLOPs of the LIVE rows only (%d of %d; the kernel pads every sample to 8 rows and every group to 32): a latency-"
            "bound launch of ~40 groups x 4 head pairs, not a roofline kernel; the plan launch (once per channel) is separate"
            % (int(live), B * T))
    else:
        add("sq_mha_core%s (L=%d text bank, masked)" % ("_bf16" if bf else "", T), (core, T, True), mfma,
            mha_core_flops(B, T, D, H, dk), "FLOPs counted over all T rows; masked row tiles are skipped, so this can exceed "
                                            "the unmasked figure")
    if bf:
        for L_, nm in ((P, "image bank"), (T, "text bank, masked")):
            add("sq_mha_folded_bf16 (L=%d %s)" % (L_, nm), ("mgnns_sq_mha_folded_bf16_fwd", L_), "hbm",
                B * (L_ * 320 * 2.0 + H * D * 4 + H * D * 2),
                "folded attention: one read of the bf16 bank [B, L, 320] + the composed query rows in, the weighted bank rows out"
                + ("; bytes counted over all T rows, masked row tiles are never read" if L_ == T else ""),
                pmc_key="folded_attn_bf16@L%d" % L_ if L_ == P else None)
        add("mha_tail_c16", ("mgnns_mha_tail_c16_fwd",), "mfma_bf16",
            B * 2.0 * (H * D * D + 2 * D * D) + B * 2.0 * D * H * D / 2,
            "the layer tail behind the folded attention: composed output map (H*D -> D) + FFN per launch; the next layer's "
            "composed query map (D -> H*D) is fused into every other launch (averaged in); bound by streaming the packed "
            "weights from L2 (1.44 MB per composed map), not by the matrix pipe",
            l2_stream_bytes=pk(D, H * D) / 4 + 2 * pk(D, D))
        add("imgbank_pool_bf16", ("mgnns_imgbank_pool_bf16_fwd",), "hbm", B * (2048.0 * P * 4 + P * D * 2 + 2048 * 4),
            "fp32 map read once + bf16 bank + pooled row written", pmc_key="imgbank_pool_bf16@B%d" % B)
        add("mha_tail_bf16", ("mgnns_mha_tail_bf16_fwd",), "mfma_bf16",
            B * 2.0 * (H * dk * D + 2 * D * D) + B * 2.0 * D * H * dk / 2,
            "fc + FFN per launch (K of fc split over a cluster of 4 workgroups, the last arriver runs LN / FFN / LN); the next "
            "layer's w_qs is a second launch behind every other tail (inside this timing): a chain of small phases on 64 "
            "CUs, bound by L2 latency / streaming, not by the matrix pipe", l2_stream_bytes=pk(D, H * dk) / 4 + 2 * pk(D, D))
    else:
        add("imgbank_pool (fp32)", ("mgnns_imgbank_pool_fwd",), "mfma_f32", B * 2.0 * P * 2048 * D, "bank projection FLOPs")
        add("mha_tail (fp32)", ("mgnns_mha_tail_fwd",), "mfma_f32",
            B * 2.0 * (H * dk * D + 2 * D * D) + B * 2.0 * D * H * dk / 2, "fc + FFN (+ w_qs on every other launch)")
    add("textgcn", ("mgnns_textgcn_fwd",), "hbm", textgcn_bytes(np.asarray(inp["text"]), cfg.ngram, D),
        "sum_b U_b*1200 + E_b*12 + 1200 B (SURVEY 8d a1)")
    lstm_key = ("mgnns_bilstm_bf16_fwd",) if (bf and os.environ.get("MGNNS_LSTM_REC", "bf16") == "bf16") else ("mgnns_bilstm_fwd",)
    add("bilstm (whole op: pack + 2 input GEMMs + 2 recurrences)", lstm_key, mfma,
        34.8e6 * B, "34.8 MFLOP/sample; sequential over <=T steps: latency-bound, not an MFMA-roofline kernel")
    for C, nm in ((cfg.C_obj, "object"), (cfg.C_place, "place")):
        add("label_gcn (%s graph, C=%d): gen_adj + 2 GraphConvolutions + w_q, one persistent launch" % (nm, C),
            ("mgnns_label_gcn_fwd", C, bf), mfma, 2.0 * C * (D * 1024 + 1024 * 2048) + 2.0 * 7 * D * D,
            "64 workgroups, three grid barriers: a latency chain (DESIGN.md section 4), not an MFMA-roofline kernel; "
            "split-bf16 operands in bf16 mode (3 MFMAs per product, not counted)")
        if bf:
            add("label_tail_bf16 (%s channel, C=%d): read-out + label attention + maps + next query, one launch" % (nm, C),
                ("mgnns_label_tail_bf16_fwd", C), "mfma_bf16",
                B * 2.0 * (2048 * C + 2 * C * D + 7 * D * 100 + 700 * D + D * H * dk),
                "4-workgroup clusters per 16 samples, split-bf16 operands (3 MFMAs per product, not counted); bound by "
                "streaming the packed weights from L2, not by the matrix pipe",
                l2_stream_bytes=2 * (pk(C, 2048) / 4 + 2 * pk(D, C) + pk(100, D) + pk(D, 700) + pk(H * dk, D) / 4))
        for F in (1024, 2048):
            nnz = int((np.asarray(getattr(model, nm + "_A").detach().cpu()) != 0).sum())
            add("spmm_csr (%s graph, F=%d)" % (nm, F), ("mgnns_spmm_csr_fwd", C, F), "hbm", nnz * 8.0 + 2.0 * C * F * 4,
                "nnz*8 + 2*C*F*4 B; the separate operator, timed beside the forward (the forward runs the label GCN as one "
                "persistent launch); model-scale graphs are launch-bound (see `stress` for the 10k-node figure)")
    return rows


# ------------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch as `python bench.py --gpus N` (self-launching) or "
                         "`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ.get("MGNNS_BENCH_TEST_KILL_RANK") == str(rank):      # tests/test_bench_launch_cpu.py: a rank that never joins
        raise SystemExit(3)
    if args.config == "stress":
        return stress_run(args, rank, local_rank, world)
    if args.dry_launch:
        return dry_run(args, rank, world)

    want_dist = world > 1 or os.environ.get("MGNNS_FORCE_DIST") == "1"   # the env switch exercises the RCCL path on one GPU
    # test hooks for a ONE-GPU box (tools/r2_check.sh): MGNNS_BENCH_BACKEND=gloo + MGNNS_BENCH_SAME_GPU=1 run every rank of
    # a multi-rank launch on cuda:0 with the gloo backend (RCCL refuses two ranks on one device), which exercises the whole
    # N-rank code path -- shards, both scalings, MAX-reduced timing, the JSON line -- except RCCL itself
    backend = os.environ.get("MGNNS_BENCH_BACKEND", "nccl")
    if os.environ.get("MGNNS_BENCH_SAME_GPU") == "1":
        local_rank = 0
    graph_collective = False
    if want_dist and backend == "nccl" and not args.no_graph and os.environ.get("MGNNS_GRAPH_COLLECTIVE", "1") == "1":
        graph_collective = probe_collective()          # child process, before this one touches the GPU

    import numpy as np
    import torch
    from mgnns_amd import _lib, harness, ops, synth
    from mgnns_amd.graph import GraphedForward, GraphedPipeline
    from mgnns_amd.sharded import ShardedForward, shard_bounds

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if want_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        ok = torch.tensor([1 if graph_collective else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)       # every rank takes the same path
        graph_collective = bool(ok.item())

    cfg = synth.CONFIGS[args.config]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    label_query = synth.make_inputs(cfg, B=1, seed=cfg.seed, pmi=pmi)["label_query"]
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, label_query, dev)
    model.set_precision({"bf16": "bf16", "bf16x3": "bf16x3"}.get(args.dtype, "fp32"))
    model.set_attention(args.attn)
    attn = model.attention                         # 'auto' resolved: folded in bf16 mode, faithful in fp32 mode
    folded_c16 = attn == "folded" and args.dtype == "bf16"      # the composed-map kernels (sq_mha_folded_bf16.hip + the c16 tail)
    if args.single_stream:
        model.use_streams = False
    core = {"bf16": "mgnns_sq_mha_core_bf16_fwd", "bf16x3": "mgnns_sq_mha_core_split_fwd"}.get(args.dtype, "mgnns_sq_mha_core_fwd")
    if attn == "folded":
        core = "mgnns_sq_mha_folded_bf16_fwd" if folded_c16 else "mgnns_sq_mha_folded_fwd"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def shard(scaling):
        """This rank's inputs: weak = its own B-sample draw; strong = its slice of the one global 256-sample batch."""
        if scaling == "weak":
            return synth.make_inputs(cfg, B=args.batch, seed=rank_seed(cfg.seed, rank, "weak"), pmi=pmi), args.batch
        lo, hi = shard_bounds(GLOBAL_BATCH, world, rank)
        g = synth.make_inputs(cfg, B=GLOBAL_BATCH, seed=rank_seed(cfg.seed, rank, "strong"), pmi=pmi)
        return {k: (v[lo:hi] if k != "label_query" else v) for k, v in g.items()}, hi - lo

    comm = None
    if dist is not None and backend == "nccl" and os.environ.get("MGNNS_COLLECTIVE", "abi") == "abi":
        try:          # the all-gather through the C ABI (mgnns_allgather_logits); torch.distributed is only the rendezvous
            from mgnns_amd.comm import AbiComm
            comm = AbiComm.from_torch_distributed(device=dev)
        except Exception as e:
            print("C-ABI communicator unavailable (%s: %s); using torch.distributed" % (type(e).__name__, e), file=sys.stderr)
        ok = torch.tensor([1 if comm is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not bool(ok.item()):
            comm = None

    def measure(scaling):
        inp, b_local = shard(scaling)
        if scaling == "strong" and b_local * world != GLOBAL_BATCH:
            raise SystemExit("strong scaling needs the global batch %d divisible by %d ranks" % (GLOBAL_BATCH, world))
        call = harness.call_args(inp, dev)
        launch = "eager"
        pipe = None
        if args.no_graph:
            fwd = ShardedForward(lambda *a: model(*a), comm=comm)
        else:
            sf = ShardedForward(lambda *a: gf.replay(), comm=comm)
            gf, fwd = None, None
            depth = max(1, args.in_flight) if getattr(model, "use_streams", True) else 1
            if dist is not None and graph_collective:
                try:                                  # the logits all-gather as a node of the same graph
                    caps = [GraphedForward(model, call, post=sf.gather, settle=False, mode="segments" if depth > 1 else None)
                            for _ in range(depth)]
                except Exception as e:                # capture of the collective refused: gather right behind the replay
                    print("collective capture failed (%s: %s); gathering after the replay" % (type(e).__name__, e), file=sys.stderr)
                    caps = None
                # every rank takes the SAME path: a rank whose capture failed would otherwise issue no all-gathers in the
                # settle / warm-up replays the others run, and the RCCL sequences would never match again
                ok = torch.tensor([0 if caps is None else 1], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()):
                    for c in caps:
                        c.settle()
                    gf = caps[0]
                    fwd = lambda *a: gf.replay()
                    launch = "hipGraph replay (RCCL all-gather captured)"
                    if depth > 1:
                        pipe = GraphedPipeline.of(caps)
                else:
                    gf = None
            if gf is None:
                gf = GraphedForward(model, call, mode="segments" if (depth > 1 and dist is None) else None)      # inputs are resident in the graph's static buffers
                fwd = sf
                launch = "hipGraph replay" + ((" + %s all-gather behind it" % ("RCCL" if backend == "nccl" else backend)) if dist is not None else "")
                if depth > 1 and dist is None:        # (a gather BEHIND the replay joins the streams anyway: one capture then)
                    pipe = GraphedPipeline.of([gf] + [GraphedForward(model, call, mode="segments") for _ in range(depth - 1)])
            launch += " [%s%s]" % (gf.mode, "; auto timed %s ms" % gf.pick_ms if gf.pick_ms else "")
            if pipe is not None:
                launch += " x %d in flight (own buffers, round robin, no join)" % depth
        out = {}

        def step_serial():
            out["logits"] = fwd(*call)

        def step():
            if pipe is None:
                return step_serial()
            out["logits"] = pipe.replay().static_out

        def regions(fn, after_warmup=None):
            """args.regions timed regions of EXACTLY args.steps steps (each bracketed by barrier + synchronize, max over ranks),
            one warm-up in front of the first -> (median region, its per-rank times, every region's ms per step)"""
            rs = []
            for r in range(max(1, args.regions)):
                d = timed_steps(fn, args.steps, args.warmup if r == 0 else 0, barrier, after_warmup if r == 0 else None)
                rs.append(max_over_ranks(d, dist, dev))
            order = sorted(range(len(rs)), key=lambda i: rs[i][0])
            med = rs[order[(len(rs) - 1) // 2]]                   # (lower) median: a region that was actually measured
            return med[0], med[1], [round(d / args.steps * 1e3, 4) for d, _ in rs]

        serial = None
        total = world * b_local
        with torch.no_grad():
            if pipe is not None:
                # the same forward one at a time (every replay joins the four streams before the next starts): reported next
                # to the headline as `serial_replay`
                ds, _, sreg = regions(step_serial)
                serial = {"ms_per_step": round(ds / args.steps * 1e3, 4), "value": round(total / (ds / args.steps), 1),
                          "unit": "samples/s", "what": "one forward in flight: a join of the four streams between replays",
                          "regions_ms_per_step": sreg}
            # after the warm-up: RCCL's banner out of every rank's C stdio buffer, long before the JSON line
            dt, dts, reg = regions(step, flush_c_stdio if dist is not None else None)
        torch.cuda.synchronize()
        _lib.take_status()                  # a persistent launch inside a replayed graph that gave up a bounded wait: no number then
        return {"inp": inp, "call": call, "b_local": b_local, "dt": dt, "dt_ranks": dts, "launch": launch,
                "value": total / (dt / args.steps), "ms": dt / args.steps * 1e3, "global_batch": total, "regions_ms": reg,
                "logits": out["logits"], "serial": serial, "in_flight": 1 if pipe is None else len(pipe.items)}

    if os.environ.get("MGNNS_BENCH_ORDER") == "strong_first" and world > 1:      # test hook: order effects
        strong = measure("strong")
        weak = measure("weak")
    else:
        weak = measure("weak")
        strong = measure("strong") if (world > 1 or os.environ.get("MGNNS_BENCH_FORCE_STRONG") == "1") else None

    # ---- roofline leg: every C-ABI launch timed with HIP events on the stream it runs on, over eager single-stream
    #      forwards right after the timed region (events cannot sit inside a graph); rank 0's shard ----
    B, inp, call = weak["b_local"], weak["inp"], weak["call"]
    if args.dtype == "bf16":          # the other attention formulation is timed below too: build its weight packs / first launches untimed
        model.use_streams = False
        model.set_attention("faithful" if attn == "folded" else "folded")
        with torch.no_grad():
            for _ in range(2):
                model(*call)
        model.set_attention(args.attn)
        torch.cuda.synchronize()
    timer = ops.KernelTimer(None)
    ops.set_timer(timer)
    model.use_streams = False
    with torch.no_grad():
        for _ in range(min(args.steps, 10)):
            model(*call)
        for _ in range(3):          # the label GCN as separate operators (gen_adj / GEMM / CSR SpMM): their own roofline rows
            model._label_gcn(model.object_A, call[5])
            model._label_gcn(model.place_A, call[6])
        other = "faithful" if attn == "folded" else "folded"       # the other formulation's kernels too: their rows of roofline_all
        if args.dtype == "bf16":
            model.set_attention(other)
            for _ in range(min(args.steps, 5)):
                model(*call)
            model.set_attention(args.attn)
    torch.cuda.synchronize()
    model.use_streams = not args.single_stream
    ops.set_timer(None)

    variants = {}
    trunks = stress = textpipe = small = None
    single = world == 1 and dist is None
    if single and not args.no_variants and not args.no_graph:
        with torch.no_grad():
            if attn == "faithful":
                model.set_attention("folded")
                variants["attention=folded"] = graphed_variant(
                    model, call, B, args.steps, args.warmup,
                    "same step, fusion attention with the projections folded away algebraically (model.set_attention('folded'); bf16 "
                    "mode: csrc/sq_mha_folded_bf16.hip + the c16 tail; fp32 mode: csrc/sq_mha_folded.hip): one read of the memory "
                    "bank per layer instead of the K/V projection GEMMs, same results to rounding (tests/: the reference's goldens); "
                    "not the formulation the MFMA target is quoted on, hence a variant", in_flight=max(1, args.in_flight))
                sb = {}
                for bs in (128, 64, 32):       # the per-GPU shards of a strong-scaling run with this attention (one forward at a time)
                    sub = {k: (v[:bs] if k != "label_query" else v) for k, v in inp.items()}
                    sb["B=%d" % bs] = graphed_variant(model, harness.call_args(sub, dev), bs, args.steps, args.warmup, "")["ms_per_step"]
                variants["attention=folded"]["small_batch_ms_per_step"] = sb
            else:
                model.set_attention("faithful")
                variants["attention=faithful (the north-star's MFMA formulation)"] = graphed_variant(
                    model, call, B, args.steps, args.warmup,
                    "same step with K and V projected from the memory bank as the reference does (csrc/sq_mha_bf16.hip: the kernel "
                    "the >= 40 % MFMA target is quoted on; its rows in roofline_all)", in_flight=max(1, args.in_flight))
            model.set_attention(args.attn)
            if args.dtype == "bf16" and ops.LSTM_FOLD_EMBEDDING:
                # the headline reads the BiLSTM's layer-0 input projection out of a table folded from the embedding and W_ih once per
                # weight version (weights only: DESIGN 3); the same step with the projection GEMM on the gathered rows in every forward
                ops.LSTM_FOLD_EMBEDDING = False
                try:
                    variants["lstm layer-0 projection computed per forward (MGNNS_LSTM_FOLD=0)"] = graphed_variant(
                        model, call, B, args.steps, args.warmup,
                        "same step, the BiLSTM's first input projection as a bf16 GEMM on the gathered embedding rows instead of rows of "
                        "the table emb . W_ih^T + b_ih (computed once per weight version; the text bank is bit-identical either way: "
                        "tests/test_ops_gpu.py)", in_flight=max(1, args.in_flight))
                finally:
                    ops.LSTM_FOLD_EMBEDDING = True
            if args.dtype == "bf16":
                model.set_precision("fp32")
                variants["dtype=f32 (parity-grade)"] = graphed_variant(
                    model, call, B, max(5, args.steps // 2), 2,
                    "same step in fp32 mode: every contraction on the exact-f32 MFMA -- the mode the north-star's 1e-4 "
                    "logit tolerance is gated on (tests/test_model_gpu.py)")
                model.set_precision("bf16x3").set_attention("folded")
                variants["dtype=bf16x3 + folded attention (parity-grade)"] = graphed_variant(
                    model, call, B, args.steps, args.warmup,
                    "same step with every heavy product in split-bf16 (bf16 hi + lo operands, three bf16 MFMAs per product, fp32 "
                    "accumulation: image banks, label GCN, channel and layer tails), exact fp32 LSTM and the exact-fp32 folded "
                    "attention: inside the 1e-4 logit gate (tests/test_model_gpu.py) without the exact-f32 MFMA's 1/16 rate")
                model.set_precision("bf16x3").set_attention("faithful")
                variants["dtype=bf16x3 + faithful attention (parity-grade, the reference's formulation)"] = graphed_variant(
                    model, call, B, args.steps, args.warmup,
                    "same step, split-bf16 everywhere as above but K and V PROJECTED from the memory bank as the reference does "
                    "(models/submodules.py:55-119) on the split-bf16 attention core (csrc/sq_mha_split_bf16.hip: hi + lo operands, "
                    "three bf16 MFMAs per product = 3 x the 61.86 GFLOP of a launch executed, bank rows in two halves joined by an "
                    "online-softmax merge): the reference's own formulation inside the 1e-4 logit gate "
                    "(tests/test_model_gpu.py::test_bf16x3_mode_stays_inside_the_parity_gate[*-faithful])",
                    in_flight=max(1, args.in_flight))
                model.set_precision("bf16").set_attention(args.attn)
        # the per-GPU shards of a strong-scaling run (global batch 256 over 2 / 4 / 8 GPUs), measured here on one GPU: what the
        # 1 -> 8 curve of configs[3] is bounded by while no multi-GPU node has run it
        small = {"what": "ms per forward of a 128 / 64 / 32-sample shard on ONE GPU (hipGraph replay, one forward at a time): global "
                         "batch 256 over 2 / 4 / 8 GPUs takes at least this long per step, i.e. strong scaling <= 256-sample ms / "
                         "shard ms (both one at a time); *_in_flight2: the same shards with two forwards in flight like the headline "
                         "(successive global batches overlap on a GPU), bound = headline ms / that",
                 "schedule": {"B=%d" % bs: model.resolve_schedule(bs) for bs in (128, 64, 32)}}
        one_at_a_time_ms = (weak["serial"] or {}).get("ms_per_step", weak["ms"])
        with torch.no_grad():
            for bs in (128, 64, 32):
                sub = {k: (v[:bs] if k != "label_query" else v) for k, v in inp.items()}
                r = graphed_variant(model, harness.call_args(sub, dev), bs, args.steps, args.warmup, "",
                                    in_flight=max(1, args.in_flight) if getattr(model, "use_streams", True) else 1)
                one = r.get("serial_replay", r)
                small["B=%d" % bs] = {"ms_per_step": one["ms_per_step"], "samples_per_s": one["value"],
                                      "strong_scaling_bound_x": round(one_at_a_time_ms / one["ms_per_step"], 2)}
                if "serial_replay" in r:
                    small["B=%d" % bs].update({"ms_per_step_in_flight2": r["ms_per_step"], "samples_per_s_in_flight2": r["value"],
                                               "strong_scaling_bound_x_in_flight2": round(weak["ms"] / r["ms_per_step"], 2)})
        stress = stress_leg(dev)
        textpipe = text_pipeline_leg(dev)
        trunks = trunk_leg(dev)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (largest share of the forward's CU time) ----
    P = inp["object_feature"].shape[2] * inp["object_feature"].shape[3]
    roofline = None
    if folded_c16:
        # folded attention: the K/V projection GEMMs are gone and the image-bank kernel (2 launches x 256 CUs x ~90 us of a
        # ~0.6 ms forward) is the largest block; HBM-bound on one read of the fp32 feature map
        durs = timer.durations_ms().get(("mgnns_imgbank_pool_bf16_fwd",), [])
        if durs:
            avg_ms = float(np.mean(durs))
            by = B * (2048.0 * P * 4 + P * cfg.emb_size * 2 + 2048 * 4)
            roofline = {"bound": "hbm", "kernel": "imgbank_pool_bf16_kernel (B=%d, %d positions x 2048 -> 300)" % (B, P),
                        "achieved": round(by / (avg_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(by / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), "traffic": None,
                        "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(durs), "bytes_per_launch": by}
            if B == GLOBAL_BATCH:
                tr, src = pmc_traffic("imgbank_pool_bf16@B%d" % B)
                roofline["traffic"] = tr
                if src:
                    roofline["traffic_source"] = src
    elif attn == "folded":
        # three launches per call (U = qh.Wk, the bank pass, o = C.Wv^T); the bank pass is HBM-bound on one read of the bank
        is_bf16 = args.dtype == "bf16"
        durs = timer.durations_ms().get((core, P, is_bf16), [])
        if durs:
            avg_ms = float(np.mean(durs))
            by = B * P * (320 * 2 if is_bf16 else cfg.emb_size * 4)
            roofline = {"bound": "hbm", "kernel": "mgnns_sq_mha_folded_fwd (3 launches, L=%d, H=%d)" % (P, cfg.n_head),
                        "achieved": round(by / (avg_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(by / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), "traffic": None,
                        "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(durs), "bytes_per_launch": by}
    else:
        durs = timer.durations_ms().get((core, P, False), [])
        if durs:
            avg_ms = float(np.mean(durs))
            fl = mha_core_flops(B, P, cfg.emb_size, cfg.n_head, cfg.d_kv)
            ach = fl / (avg_ms * 1e-3) / 1e12
            kname = {"bf16": "sq_mha_core_bf16_kernel", "bf16x3": "sq_mha_core_split_kernel"}.get(args.dtype, "sq_mha_core_kernel")
            roofline = {"bound": "mfma", "kernel": "%s (L=%d, H=%d)" % (kname, P, cfg.n_head),
                        "achieved": round(ach, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                        "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(durs),
                        "flops_per_launch": fl}
            if B == GLOBAL_BATCH:
                tr, src = pmc_traffic("%s@L%d" % (kname, P))
                roofline["traffic"] = tr
                if src:
                    roofline["traffic_source"] = src
            # the same kernel as 40 launches back to back in one hipGraph between ONE event pair (no per-launch event overhead: the
            # per-launch figure above reads ~4 % over rocprofv3's kernel duration), and the committed rocprofv3 average next to it
            b2b = core_back_to_back(model, cfg, B, P, args.dtype)
            if b2b:
                roofline["avg_launch_ms_back_to_back"] = round(b2b, 4)
                roofline["frac_back_to_back"] = round(fl / (b2b * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4)
            rp_us, rp_src = rocprof_avg_us(kname, {"bf16": "bf16_serial", "bf16x3": "bf16x3_serial"}.get(args.dtype, "f32"))
            if rp_us:
                roofline["avg_launch_ms_rocprof"] = round(rp_us * 1e-3, 4)
                roofline["frac_rocprof"] = round(fl / (rp_us * 1e-6) / 1e12 / PEAK_TFLOPS[args.dtype], 4)
                roofline["rocprof_source"] = rp_src + " (committed profile: another box of the pool)"
            if args.dtype == "bf16x3":
                roofline["executed_flops_per_launch"] = 3 * fl
                roofline["note"] = ("split-bf16: three MFMAs per product -- the pipe executes 3 x flops_per_launch; achieved / frac are "
                                    "quoted on the ALGORITHMIC flops of the reference's formulation (unchanged)")
    rall = roofline_all(timer, cfg, B, P, inp, args.dtype, model)

    cpu = None
    parity = None
    if not args.no_cpu_baseline and single:          # N = 1 only (rank 0's host cores)
        ref, cpu = cpu_baseline(cfg, model, inp, pmi)
        parity = float((weak["logits"][:B].float().cpu() - ref).abs().max())
        for v in variants.values():
            v["max_abs_logit_diff_vs_cpu_oracle"] = float((v["_out"] - ref).abs().max())
        if not args.no_variants:
            try:
                cpu["other_configs"] = cpu_baseline_other()
            except Exception as e:
                cpu["other_configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
    for v in variants.values():
        del v["_out"]

    head = weak if (args.scaling == "weak" or strong is None) else strong
    line = {
        "metric": "forward samples/sec at batch 256 (3-channel GCN + fusion)",
        "value": round(head["value"], 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(head["ms"], 4),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s: B=%d/GPU, T=%d, V=%d, %d heads x %d layers, C=(%d,%d), fp32 feature maps [B,2048,14,14] "
                               "resident in HBM"
                               % (cfg.name, head["b_local"], cfg.T, cfg.V, cfg.n_head, cfg.stack_num, cfg.C_obj, cfg.C_place),
                   "global_batch": head["global_batch"], "parallelism": "batch-shard x%d" % world,
                   "launch": head["launch"],
                   "attention": attn + (" (K/V and query/output projections composed into the maps either side of the attention: "
                                        "one read of the memory bank per layer; the reference's explicit formulation is "
                                        "variants['attention=faithful ...'])" if folded_c16 else ""),
                   "collective": (None if dist is None else "RCCL all-gather via the C ABI (mgnns_allgather_logits)" if comm is not None
                                  else "RCCL all-gather via torch.distributed" if backend == "nccl" else backend)},
        "roofline": roofline, "cpu_baseline": cpu, "max_abs_logit_diff_vs_cpu_oracle": parity,
    }
    line["config"]["forwards_in_flight"] = head.get("in_flight", 1)
    built, tree, same = _lib.source_state()          # which sources the library that ran was built from (mgnns_amd/build.py)
    line["config"]["sources"] = built if same else "%s (the tree next to it: %s)" % (built, tree)
    if hasattr(model, "resolve_schedule"):      # segment -> stream schedule of the timed forward (mgnns_amd/model.py::SCHEDULES)
        line["config"]["schedule"] = model.resolve_schedule(B)
    if args.dtype == "bf16":
        line["config"]["lstm_layer0_projection"] = ("rows of the table emb . W_ih^T + b_ih folded once per weight version (weights only)"
                                                    if ops.LSTM_FOLD_EMBEDDING else "bf16 GEMM on the gathered rows, every forward")
    try:            # the placement assumption of the slab / row-range kernels (speed only), measured on this device
        ok, ids = _lib.xcd_probe()
        line["config"]["xcd_map"] = {"blockIdx_and_7_selects_the_xcd": ok, "xcc_id_per_residue": ids}
    except Exception as e:
        line["config"]["xcd_map"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # `steps` steps per region; ms_per_step / value = the median of these regions (one warm-up in front of the first)
    line["timing"] = {"regions": len(head["regions_ms"]), "steps_per_region": args.steps, "statistic": "median region",
                      "regions_ms_per_step": head["regions_ms"], "min_ms_per_step": min(head["regions_ms"]),
                      "max_ms_per_step": max(head["regions_ms"]),
                      "value_is": ("throughput with %d forwards in flight (captures with buffers of their own, no join between "
                                   "replays); one forward at a time = serial_replay" % head.get("in_flight", 1))
                      if head.get("in_flight", 1) > 1 else "one forward at a time"}
    if head.get("serial"):
        line["serial_replay"] = head["serial"]
        # the same figures as top-level scalars (a parser that keeps only scalar keys keeps them)
        line["value_one_in_flight"] = head["serial"]["value"]
        line["ms_per_step_one_in_flight"] = head["serial"]["ms_per_step"]
    for key, vname, field in (("value_without_lstm_fold", "lstm layer-0 projection computed per forward (MGNNS_LSTM_FOLD=0)", "value"),
                              ("value_fp32_parity_mode", "dtype=f32 (parity-grade)", "value"),
                              ("value_bf16x3_faithful", "dtype=bf16x3 + faithful attention (parity-grade, the reference's formulation)", "value"),
                              ("value_bf16x3_folded", "dtype=bf16x3 + folded attention (parity-grade)", "value"),
                              ("value_folded_attention", "attention=folded", "value")):
        if vname in variants:
            line[key] = variants[vname][field]
            if "max_abs_logit_diff_vs_cpu_oracle" in variants[vname]:
                line[key.replace("value_", "max_abs_logit_diff_") ] = variants[vname]["max_abs_logit_diff_vs_cpu_oracle"]
    if strong is not None:
        for nm, r in (("weak_scaling", weak), ("strong_scaling", strong)):
            line[nm] = {"value": round(r["value"], 1), "unit": "samples/s", "ms_per_step": round(r["ms"], 4),
                        "per_gpu_batch": r["b_local"], "global_batch": r["global_batch"], "launch": r["launch"],
                        "dt_ranks_s": [round(t, 6) for t in r["dt_ranks"]]}
    if rall:
        line["roofline_all"] = rall
    if variants:
        line["variants"] = variants
    if small is not None:
        line["small_batch"] = small
    if stress is not None:
        line["stress"] = stress
    if textpipe is not None:
        line["text_pipeline"] = textpipe
    if trunks is not None:
        line["trunks"] = trunks
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line must be the last thing on stdout: push out whatever native libraries (RCCL banner) still hold in
    # the C stdio buffer first
    assert line["n_gpus"] == args.gpus
    emit(line)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.probe_collective:
        return probe_collective_child()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
