#!/usr/bin/env python3
"""GPU-box tool: end-to-end vg_scan_topk latency (host query in -> host top-k out) on small corpora.
    python tools/tools_latency.py [--rows 10000] [--dim 384]
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=str, default="1000,10000,100000,1000000")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--reps", type=int, default=2000)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    rng = np.random.default_rng(42)
    for n in [int(x) for x in args.rows.split(",")]:
        rows = rng.standard_normal((n, args.dim), dtype=np.float32)
        c = pkg.Corpus(pkg.F32, args.dim, capacity=n)
        c.append(rows)
        qs = rng.standard_normal((64, args.dim), dtype=np.float32)
        for i in range(20):
            c.scan_topk(pkg.L2, qs[i % 64], 20)
        lat = np.empty(args.reps)
        for i in range(args.reps):
            t0 = time.perf_counter()
            c.scan_topk(pkg.L2, qs[i % 64], 20)
            lat[i] = time.perf_counter() - t0
        c.set_profiling(True)
        for i in range(50):
            c.scan_topk(pkg.L2, qs[i % 64], 20)
        _, sm, mm = c.profile_mean_ms()
        c.set_profiling(False)
        print("rows %8d dim %d: p50 %.1f us  min %.1f us  p99 %.1f us   (events: scan %.1f us merge %.1f us)  VG_HOST_DIRECT=%s"
              % (n, args.dim, np.percentile(lat, 50) * 1e6, lat.min() * 1e6, np.percentile(lat, 99) * 1e6, sm * 1e3, mm * 1e3,
                 os.environ.get("VG_HOST_DIRECT", "auto")), flush=True)
        c.close()

if __name__ == "__main__":
    main()
