"""CPU-only: the committed default bench line (profiles/*_bench_default_line.json, the output of `python bench.py` on one MI355X)
carries every key the driver's contract names, and its numbers are consistent with each other the way a reviewer would re-derive
them: value = rows / step time, roofline.achieved = algorithmic bytes / kernel time, frac = achieved / peak, kernel time <= step
time, PMC traffic within a percent of the algorithmic bytes and taken on THESE kernel sources."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default_line.json")))
    if not files:
        pytest.skip("no committed bench line")
    return files[-1], json.load(open(files[-1]))


def test_default_line_has_the_contract_keys_and_adds_up():
    path, d = latest_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, (path, key)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    rows = d["config"]["rows_per_gpu"]
    algo = rows * d["config"]["dim"] * 4
    assert algo == r["algorithmic_bytes_per_launch"] == 15_360_000_000
    assert abs(r["achieved"] - algo / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.5 < r["frac"] < 1.0
    assert r["kernel_ms"] < d["ms_per_step"] < r["kernel_ms"] + 0.2                      # one kernel + the per-query plumbing
    assert abs(d["value"] - rows / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert r["traffic"] is not None and abs(r["traffic"] / algo - 1.0) < 0.01          # no wasted re-reads
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["unit"] == d["unit"] and c["value"] < d["value"] / 50


def test_also_lines_add_up_too():
    path, d = latest_line()
    a = d["also"]
    c3 = a["c3"]["roofline"]
    assert c3["algorithmic_bytes_per_launch"] == 7_680_000_000 and abs(c3["frac"] - 7.68e9 / (c3["kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-6
    c5 = a["c5"]["roofline"]
    assert c5["bound"] == "mfma" and c5["flops_per_launch"] == 2.0 * 1024 * 10_000_000 * 384
    assert abs(c5["achieved"] - c5["flops_per_launch"] / (c5["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * c5["achieved"] and c5["achieved"] < c5["peak"]
    c4 = a["c4_one_gpu"]["roofline"]
    assert c4["algorithmic_bytes_per_launch"] == 153_600_000_000 and 0.7 <= c4["frac"] < 1.0        # north_star's target on its named shape
    for row in a["kernel_matrix"]["rows"]:
        assert abs(row["frac"] - row["algorithmic_bytes_per_launch"] / (row["kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-6
    f = d["filter_scan"]
    assert f["last_query_same_answer_as_plain_scan"] is True and f["kernel_ms"] < d["roofline"]["kernel_ms"]


def test_pmc_traffic_file_matches_the_kernel_sources():
    """bench.py only copies a PMC figure measured on the kernel sources it runs: the committed file must be current"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    hashes = {v["kernel_source_hash"] for k, v in t.items() if isinstance(v, dict)}
    assert len(hashes) == 1                                      # one PMC pass, one build
    if hashes != {bench.kernel_source_hash()}:
        # not an error by itself - bench.py then reports traffic: null with the reason - but the round should not end like this
        pytest.skip("profiles/pmc_traffic.json was measured on other kernel sources: re-run `tools/measure.sh <tag> pmc` and copy it")
