#!/usr/bin/env python3
"""The plain f32 scan at two corpus sizes, same kernel, same launch shape: why does 100M x 384 stream 4-5 % slower than 10M x 384
?  Builds N x 384 f32 on the device, runs `--scans` plain-kernel top-20 scans and prints the kernel's mean
milliseconds / TB/s from the engine's own HIP events.  Run it bare, and under `rocprofv3 --pmc <counters>` (tools/measure.sh tlb)
for the address-translation and wait counters of the same dispatches.  VG_SCAN_ORDER=0 / 1 picks the batch order."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--scans", type=int, default=12)
    ap.add_argument("--orders", default="", help="comma list of VG_SCAN_ORDER values to run one after the other in this process")
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    import bench
    pkg = g.load_package()
    c = bench.make_shard(pkg, torch, pkg.F32, a.dim, a.rows, 7, 0)
    c.set_scan_filter(0)
    rng = np.random.default_rng(43)
    qs = rng.standard_normal((a.scans + 2, a.dim), dtype=np.float32)
    for order in ([None] if not a.orders else a.orders.split(",")):
        if order is not None:
            os.environ["VG_SCAN_ORDER"] = order
        c.scan_topk(pkg.L2, qs[0], 20)
        c.scan_topk(pkg.L2, qs[1], 20)
        c.set_profiling(True)
        for i in range(a.scans):
            c.scan_topk(pkg.L2, qs[2 + i], 20)
        n, ms, merge_ms, _ = c.profile_mean_ms_ex()
        c.set_profiling(False)
        gb = a.rows * a.dim * 4 / 1e9
        print(json.dumps({"rows": a.rows, "dim": a.dim, "order": os.environ.get("VG_SCAN_ORDER", "default"), "kernel": c.kernel_name(pkg.L2), "launches": n,
                          "kernel_ms": round(ms, 4), "tb_s": round(gb / ms, 4), "frac_of_8tb": round(gb / ms / 8.0, 4)}), flush=True)
    c.close()


if __name__ == "__main__":
    main()
