"""bench.py's launch contract on a CPU-only box (`python bench.py --gpus 8` used to run ONE rank and print
n_gpus: 1).  `--gpus N` without a rank environment must start N ranks itself, and must refuse - loudly, non-zero - to print a
line when it cannot have N devices or when the launcher's WORLD_SIZE disagrees with --gpus."""
import json
import os
import subprocess
import sys

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
