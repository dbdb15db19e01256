#!/usr/bin/env python3
"""GPU-box tool: what a query costs OUTSIDE its scan kernel (the per-query floor of VERDICT r2 #9), per switch setting.
For C2 (10M x 384 f32 L2 top-20) and C3 (10M x 768 uint8 cosine top-10), default (filter) path and plain path:
wall-clock ms per query with the profiling events off (what a caller pays) and on (what bench.py's timed region pays),
and the mean kernel / pre-pass / merge milliseconds of the same queries.  One JSON line per (workload, path, setting).
    python tools/floor_ab.py [--rows 10000000] [--queries 200]
Scratch measurement aid (not part of the product or of bench.py's contract).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETTINGS = [
    ("default", {}),
    ("keys_copy", {"VG_KEYS_DIRECT": "0"}),
    ("mirror_copy", {"VG_SCAN_FILTER_MIRROR_COPY": "1"}),
    ("round3_form", {"VG_KEYS_DIRECT": "0", "VG_SCAN_FILTER_MIRROR_COPY": "1"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    import bench
    pkg = g.load_package()
    for wl, vt, dim, metric, k in (("c2", pkg.F32, 384, pkg.L2, 20), ("c3", pkg.U8, 768, pkg.COSINE, 10)):
        corpus = pkg.Corpus(vt, dim, capacity=args.rows)
        blk = 1_000_000
        gen = torch.Generator(device="cuda"); gen.manual_seed(7)
        for r0 in range(0, args.rows, blk):
            nr = min(blk, args.rows - r0)
            if vt == pkg.F32:
                t = torch.rand((nr, dim), device="cuda", dtype=torch.float32, generator=gen)
            else:
                t = (torch.rand((nr, dim), device="cuda", dtype=torch.float32, generator=gen) * 255 + 0.5).to(torch.uint8)
            torch.cuda.synchronize()
            corpus.append_device(t.data_ptr(), nr, dim * pkg.TYPE_SIZE[vt])
            del t
        rng = np.random.default_rng(3)
        if vt == pkg.F32:
            qs = rng.random((args.queries, dim), dtype=np.float32)
        else:
            qs = np.floor(rng.random((args.queries, dim)) * 255 + 0.5).astype(np.uint8)
        for path, mode in (("filter", 1), ("plain", 0)):
            corpus.set_scan_filter(mode)
            for q in qs[:12]:
                corpus.scan_topk(metric, q, k)               # shadow copy, probe, code objects: not timed
            for rep in range(args.reps):
                for name, env in SETTINGS:
                    if path == "plain" and "MIRROR" in "".join(env) and "VG_KEYS_DIRECT" not in env:
                        continue                             # (the counter mirror belongs to the filter scans)
                    saved = {kk: os.environ.get(kk) for kk in env}
                    os.environ.update(env)
                    out = {"workload": wl, "path": path, "setting": name, "rep": rep}
                    for prof in (0, 1):
                        corpus.set_profiling(bool(prof))
                        for q in qs[:5]:
                            corpus.scan_topk(metric, q, k)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for q in qs:
                            corpus.scan_topk(metric, q, k)
                        el = (time.perf_counter() - t0) / len(qs) * 1e3
                        out["ms_per_query_profiling_%s" % ("on" if prof else "off")] = round(el, 4)
                        if prof:
                            n, scan_ms, merge_ms, pre_ms = corpus.profile_mean_ms_ex()
                            out.update({"kernel": corpus.kernel_name(metric), "kernel_ms": round(scan_ms, 4), "prepass_ms": round(pre_ms, 4),
                                        "merge_ms": round(merge_ms, 4), "outside_kernel_us_profiling_on": round((el - scan_ms) * 1e3, 1)})
                    out["outside_kernel_us_profiling_off"] = round((out["ms_per_query_profiling_off"] - out["kernel_ms"]) * 1e3, 1)
                    corpus.set_profiling(False)
                    for kk, v in saved.items():
                        if v is None:
                            os.environ.pop(kk, None)
                        else:
                            os.environ[kk] = v
                    print(json.dumps(out), flush=True)
        del corpus
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
