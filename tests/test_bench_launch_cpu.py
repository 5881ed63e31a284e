"""bench.py's N-rank protocol without a GPU (`--dry-launch`: gloo, no model): the self-launcher starts N ranks before any
GPU call and fails unless all join, ranks are seeded per scaling mode, the timed region is bracketed by barriers and the
time is the MAX over ranks, the logits all-gather has the right extent, and stdout carries exactly ONE JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def _check_line(out, n, batch, strict=True):
    lines = [ln for ln in out.splitlines() if ln.strip()]
    if strict:
        assert len(lines) == 1, "stdout must hold exactly one line, got %r" % lines
    else:       # under torchrun every rank's stdout is forwarded: gloo's connection banner may precede the line
        assert sum(ln.startswith("{") for ln in lines) == 1 and lines[-1].startswith("{"), lines
    line = json.loads(lines[-1])
    assert line["n_gpus"] == n and line["dry_launch"] is True and line["scaling"] == "weak"
    assert line["steps"] == 5 and line["warmup"] == 1
    w, s = line["weak_scaling"], line["strong_scaling"]
    # rank seeding: weak = a different shard per rank, strong = the same global batch on every rank (sliced)
    assert w["seeds"] == [1237 + 1000 * r for r in range(n)]
    assert s["seeds"] == [1237] * n
    # the step of rank r sleeps (r+1) ms: the reported time is the slowest rank's, and every rank left the closing barrier
    # together (all local times within a step of the maximum)
    for r in (w, s):
        assert len(r["dt_ranks"]) == n
        assert r["ms_per_step"] >= 1.0 * n
        assert r["ms_per_step"] == pytest.approx(max(r["dt_ranks"]) / 5 * 1e3)
        assert max(r["dt_ranks"]) - min(r["dt_ranks"]) < 0.05
    assert w["gathered_rows"] == n * batch and s["gathered_rows"] == 256
    assert line["value"] == pytest.approx(n * batch / (w["ms_per_step"] * 1e-3), rel=1e-3)
    assert s["value"] == pytest.approx(256 / (s["ms_per_step"] * 1e-3), rel=1e-3)
    return line


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--steps", "5", "--warmup", "1", "--batch", "64"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_line(r.stdout, 2, 64)


def test_strong_headline_and_single_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--steps", "5", "--warmup", "1", "--scaling", "strong"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line["scaling"] == "strong" and line["value"] == pytest.approx(line["strong_scaling"]["value"], rel=1e-3)
    r = subprocess.run([sys.executable, BENCH, "--dry-launch", "--steps", "5", "--warmup", "1"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line["n_gpus"] == 1 and line["weak_scaling"]["gathered_rows"] == 256


def test_under_torchrun_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry-launch", "--steps", "5",
                        "--warmup", "1", "--batch", "32"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_line(r.stdout, 2, 32, strict=False)


def test_world_size_mismatch_fails_loudly():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_a_dead_rank_fails_the_launch(tmp_path):
    # rank 1 dies at start-up (bad --config is only read by real runs, so break it through the environment instead)
    env = _env()
    env["MGNNS_BENCH_TEST_KILL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "0", "--launch-timeout", "60"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "launch failed" in r.stderr


@pytest.mark.parametrize("n", [1, 2, 4])
def test_stress_config_shards_channels_then_batch(n):
    """`--config stress` (BASELINE configs[4]) through the same launcher: channels are dealt to ranks, beyond three ranks a
    channel's read-out batch is split; rank 0's line proves every (channel, sample) is computed exactly once."""
    cmd = [sys.executable, BENCH, "--config", "stress", "--dry-launch", "--steps", "3", "--warmup", "1"] + (["--gpus", str(n)] if n > 1 else [])
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["scaling"] == "strong" and line["dry_launch"] is True
    assert len(line["shards"]) == n and len(line["dt_ranks"]) == n
    seen = set()
    for rank_shards, shapes in zip(line["shards"], line["block_shapes"]):
        assert rank_shards
        for (c, b0, b1), (rows, cols) in zip(rank_shards, shapes):
            assert rows == b1 - b0 and cols == 10000
            for b in range(b0, b1):
                assert (c, b) not in seen
                seen.add((c, b))
    assert len(seen) == 3 * 512
    assert line["value"] == pytest.approx(3 * 512 / (line["ms_per_step"] * 1e-3), rel=1e-3)
    assert line["ms_per_step"] == pytest.approx(max(line["dt_ranks"]) / 3 * 1e3, abs=1e-3)


# ---- the result line stays small enough for the driver to parse (round 5's 21.7 KB line was not) ----
def _full_line():
    """A synthetic full result object: round 5's committed line (every detail leg present, 21.7 KB)."""
    with open(os.path.join(ROOT, "profiles", "r05_bench_bf16.json")) as f:
        return json.load(f)


def test_compact_line_is_small_strict_json_and_keeps_the_contract():
    sys.path.insert(0, ROOT)
    import bench
    full = _full_line()
    assert len(json.dumps(full)) > 20000
    compact, detail = bench.compact_line(full)
    text = json.dumps(compact, allow_nan=False)                 # strict: no NaN / Infinity tokens
    assert len(text) <= bench.LINE_TARGET_BYTES < bench.LINE_LIMIT_BYTES <= 8000
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "max_abs_logit_diff_vs_cpu_oracle"):
        assert k in back, k
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert back["roofline"][k] == full["roofline"][k]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"]
    assert back["config"]["workload"] and "model" not in back["config"]
    for k in ("roofline_all", "variants", "small_batch", "stress", "text_pipeline", "trunks", "timing"):
        assert k not in back and k in detail                    # moved, not lost
    assert all(len(v) <= bench.STR_LIMIT for v in back["config"].values() if isinstance(v, str))
    assert back["value_bf16x3_faithful"] == full["value_bf16x3_faithful"]
    digest = json.dumps(bench.summary_of(full), allow_nan=False)
    assert len(digest) + len(text) < bench.LINE_LIMIT_BYTES


def test_emit_prints_the_compact_line_last(tmp_path, capsys, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "DETAIL_FILE", str(tmp_path / "bench_detail.json"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_full_line())
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert [ln.startswith("{") for ln in lines] == [False] * (len(lines) - 1) + [True]
    assert lines[0].startswith("bench_detail ") and lines[-2].startswith("bench_summary ")
    last = json.loads(lines[-1])
    assert len(lines[-1]) < 8000 and last["roofline"] and last["cpu_baseline"]
    with open(tmp_path / "bench_detail.json") as f:
        assert json.load(f)["roofline_all"] == json.loads(lines[0][len("bench_detail "):])["roofline_all"]
    huge = dict(_full_line())
    huge["config"] = dict(huge["config"], **{"workload": "x" * 100000})
    bench.emit(huge)                                            # over-long strings are cut, not fatal
    assert len(capsys.readouterr().out.splitlines()[-1]) < 8000
