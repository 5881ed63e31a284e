"""Two (and three) FRESH child ranks on the one GPU of the box run the real sharded path -- product model, eager launches,
mgnns_amd.sharded.ShardedForward with a gloo rendezvous -- and the gathered logits are compared with this process's own
single-process logits (models/Multi_GCN_Multihead_att.py:431-567 has no cross-sample term in eval: that is what batch sharding
relies on and what this pins).  Also BASELINE configs[4]'s shard plan with three processes.  The children are started before
they touch the GPU (subprocess of a fresh interpreter; nothing is exec'ed from an initialised process)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from mgnns_amd import harness, stress, synth
from mgnns_amd.sharded import shard_bounds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "sharded_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, argv, tmp_path, timeout=600):
    """Start `world` children, wait for all, return their output files.  A rank that dies or hangs fails the test with its log."""
    port = _free_port()
    procs, outs, logs = [], [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        out = str(tmp_path / ("rank%d.pt" % r))
        log = open(str(tmp_path / ("rank%d.log" % r)), "w")
        outs.append(out)
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, WORKER, argv[0], out] + [str(a) for a in argv[1:]], env=env, stdout=log,
                                      stderr=subprocess.STDOUT, cwd=ROOT))
    try:
        for r, p in enumerate(procs):
            try:
                rc = p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                rc = None
            if rc != 0:
                logs[r].flush()
                raise AssertionError("rank %d of %d: rc=%s\n%s" % (r, world, rc, open(logs[r].name).read()[-3000:]))
    finally:
        for p in procs:                         # (exact PIDs we started)
            if p.poll() is None:
                p.kill()
        for log in logs:
            log.close()
    return outs


@pytest.mark.parametrize("precision,world,B", [("fp32", 2, 16), ("bf16", 2, 16), ("fp32", 3, 18)])
def test_child_ranks_gathered_logits_equal_the_single_process_logits(precision, world, B, tmp_path):
    """16 / 18 samples (equal shards; ragged text lengths incl. the forced longest / shortest) over `world` ranks, eager launches.  (i) bit-equal to
    this process running the model shard by shard (same launches, same data: determinism across processes); (ii) against the
    single forward over all samples within 1e-6 (fp32) -- a sample's result may differ in the last bit with the batch around
    it (16-sample tiles, K-split arrival order), never by a cross-sample term -- resp. the bf16 mode's own batch-composition
    spread (packed masked rows share workgroups: 2e-3)."""
    cfg_name, seed = "tumemo_b64", 4321
    outs = _run_ranks(world, ["forward", cfg_name, B, seed, precision], tmp_path)
    cfg = synth.CONFIGS[cfg_name]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp = synth.make_inputs(cfg, B=B, seed=seed, pmi=pmi)
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], DEV)
    model.set_precision(precision)
    with torch.no_grad():
        whole = model(*harness.call_args(inp, DEV)).cpu()
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(B, world, r)
            sub = {k: (v[lo:hi] if k != "label_query" else v) for k, v in inp.items()}
            parts.append(model(*harness.call_args(sub, DEV)).cpu())
    by_shard = torch.cat(parts, 0)
    got = [torch.load(o) for o in outs]
    for r, g in enumerate(got):
        assert g["rank"] == r and (g["lo"], g["hi"]) == shard_bounds(B, world, r)
        assert tuple(g["logits"].shape) == (B, cfg.NL)
        assert torch.equal(g["logits"], got[0]["logits"])              # every rank holds the same gathered matrix
    assert torch.equal(got[0]["logits"], by_shard), float((got[0]["logits"] - by_shard).abs().max())
    spread = float((got[0]["logits"] - whole).abs().max())
    print("sharded (%s, %d ranks) vs one forward over the batch: max |dlogit| = %.2e" % (precision, world, spread))
    assert spread < (1e-6 if precision == "fp32" else 2e-3)


def test_child_ranks_replaying_graphs_gather_device_logits(tmp_path):
    """Two child ranks capture their slice as hipGraphs and gather the DEVICE logits of each replay over gloo, eight steps back to
    back without a host synchronisation (bench.py's timed loop).  The gathered logits equal this process's shard-by-shard
    logits bit for bit, and the steps stay short: this pattern read 20-230 ms per step until ShardedForward joined the device
    in front of a gloo collective on device tensors (tools/dev/two_proc_gloo.py reproduces it without this library)."""
    cfg_name, seed, B, world, precision = "tumemo_b64", 977, 16, 2, "bf16"
    outs = _run_ranks(world, ["forward_graph", cfg_name, B, seed, precision], tmp_path)
    cfg = synth.CONFIGS[cfg_name]
    pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
    A_obj, A_place = harness.synthetic_adjacencies(cfg)
    inp = synth.make_inputs(cfg, B=B, seed=seed, pmi=pmi)
    model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], DEV)
    model.set_precision(precision)
    parts = []
    with torch.no_grad():
        for r in range(world):
            lo, hi = shard_bounds(B, world, r)
            sub = {k: (v[lo:hi] if k != "label_query" else v) for k, v in inp.items()}
            parts.append(model(*harness.call_args(sub, DEV)).cpu())
    by_shard = torch.cat(parts, 0)
    got = [torch.load(o) for o in outs]
    for g in got:
        print("rank %d: %s graph, eight replay + gather steps: mean %.2f ms, slowest %.2f ms"
              % (g["rank"], g["graph_mode"], g["mean_step_ms"], g["worst_step_ms"]))
        assert torch.equal(g["logits"], by_shard), float((g["logits"] - by_shard).abs().max())
        assert g["mean_step_ms"] < 10.0          # (degraded: 20-230 ms in EVERY step; a single slow gloo step reads 5-20 ms)


def test_stress_shard_plan_in_three_child_processes(tmp_path):
    """BASELINE configs[4] at n = 1 500, batch 48: three processes take plan_shards' blocks (whole channels, no collective); put
    together they are the one-rank result bit for bit."""
    n, B, world = 1500, 48, 3
    outs = _run_ranks(world, ["stress", n, B], tmp_path)
    one = stress.StressWorkload(0, 1, n=n, batch=B, dev=DEV).forward()
    full = {c: v.cpu() for (c, b0, b1), v in one.items()}
    seen = {c: torch.zeros(B, dtype=torch.bool) for c in full}
    for r, o in enumerate(outs):
        blocks = torch.load(o)
        assert set(blocks) == set(stress.StressWorkload(r, world, n=n, batch=B, dev=DEV).forward())
        for (c, b0, b1), v in blocks.items():
            assert torch.equal(v, full[c][b0:b1]), (r, c, b0, b1)
            assert not seen[c][b0:b1].any()
            seen[c][b0:b1] = True
    assert all(m.all() for m in seen.values())
