"""bench.py's launch contract on a CPU-only box (`python bench.py --gpus 8` used to run ONE rank and print
n_gpus: 1).  `--gpus N` without a rank environment must start N ranks itself, and must refuse - loudly, non-zero - to print a
line when it cannot have N devices or when the launcher's WORLD_SIZE disagrees with --gpus."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_n_self_launches_n_ranks():
    """no torchrun around it: bench.py re-executes itself under torch.distributed.run; rank 0's line says n_gpus = 2
    (--selftest-launch: the exchange + merge + timing plumbing on fabricated keys over gloo - no device needed)"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launch", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["ms_per_step"] > 0
    # the contract's keys, checked on a line this test produced (not on a committed file)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "data", "config"):
        assert key in out, key
    assert out["value"] is None and "workload" in out["config"]                  # (it measures nothing and says so)


def test_gpus_n_refuses_without_n_devices():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert p.returncode != 0
    assert "only" in p.stderr and "device" in p.stderr, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]          # no line at all, certainly not an n_gpus: 1 one


def test_world_size_and_gpus_must_agree():
    env = _env()
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "must agree" in p.stderr, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_eight_ranks_on_the_visible_devices_pass_the_self_check():
    """the exact code path of the driver's `bench.py --gpus 8` (self-launch under torch.distributed.run, one shard of config C4 per
    rank = 12.5M x 384 f32 rows, candidate exchange, merge, the self-check of query 0 against the recorded one-device answer in
    tests/golden/bench_c4_expected.json, max-over-ranks timing) on however many devices this box has: VG_BENCH_SHARE_DEVICES=1 lets the
    ranks share them (gloo carries the exchange when two ranks sit on one device - RCCL refuses that).  Eight shards are 154 GB: they fit
    ONE MI355X.  A functional run - the line says so - but a wrong merged top-20 fails it the way it would fail the driver's run."""
    import torch
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(0)
    full = free > 200 * (1 << 30)
    env = _env()
    env["VG_BENCH_SHARE_DEVICES"] = "1"
    cmd = [sys.executable, BENCH, "--gpus", "8", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    if not full:                                                                  # (another process holds the device's memory: a smaller shard,
        cmd += ["--rows", "1000000"]                                              #  for which no recorded answer exists)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 4 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["roofline"]["per_rank"]) == 8 and all(r["kernel_ms"] > 0 for r in out["roofline"]["per_rank"])
    if torch.cuda.device_count() < 8:
        assert "functional check" in out["config"]["note"]
    if full:
        chk = out["self_check"]
        assert chk["rowids_match"] and chk["distance_bits_match"] and not chk.get("failed"), chk
