#!/usr/bin/env python3
"""GPU-box tool: the store-mode scan (every row's distance written out: the *_stream TVFs, k > 64, vg_scan_distances) per type and
metric - kernel time by HIP events, algorithmic bytes = N * D * elem read + 4 N written.  Scratch measurement aid.
    python tools/store_mode_bench.py [--rows 10000000] [--dim 384]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--types", type=str, default="1,2,3,4")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    names = {1: "f32", 2: "f16", 3: "bf16", 4: "u8", 5: "i8"}
    mnames = {1: "l2", 3: "cos", 4: "dot", 5: "l1"}
    dim, n = args.dim, args.rows
    for vt in [int(x) for x in args.types.split(",")]:
        es = pkg.TYPE_SIZE[vt]
        c = pkg.Corpus(vt, dim, capacity=n)
        for r0 in range(0, n, 1_000_000):
            nr = min(1_000_000, n - r0)
            if vt == 1:
                t = torch.randn((nr, dim), device="cuda", dtype=torch.float32)
            elif vt == 2:
                t = torch.randn((nr, dim), device="cuda", dtype=torch.float16)
            elif vt == 3:
                t = torch.randn((nr, dim), device="cuda", dtype=torch.bfloat16)
            else:
                t = torch.randint(0, 256, (nr, dim), device="cuda", dtype=torch.uint8)
            torch.cuda.synchronize()
            c.append_device(t.data_ptr(), nr, dim * es)
            del t
        rng = np.random.default_rng(1)
        q32 = rng.standard_normal(dim, dtype=np.float32)
        q = {1: q32, 2: q32.astype(np.float16).view(np.uint16), 3: (q32.view(np.uint32) >> 16).astype(np.uint16),
             4: rng.integers(0, 256, dim).astype(np.uint8)}[vt]
        c.set_scan_filter(0)
        line = "%-5s %d x %d store mode:" % (names[vt], n, dim)
        for m in (1, 3, 4, 5):
            c.scan_distances(m, q)
            c.set_profiling(True)
            for _ in range(args.reps):
                c.scan_distances(m, q)
            nl, scan_ms, merge_ms, pre_ms = c.profile_mean_ms_ex()
            c.set_profiling(True)
            for _ in range(args.reps):
                c.scan_topk(m, q, 20)
            nl2, topk_ms, _, _ = c.profile_mean_ms_ex()
            line += "  %s %.3f ms %5.0f GB/s (top-k %.3f ms)" % (mnames[m], scan_ms, (n * dim * es + 4 * n) / (scan_ms * 1e-3) / 1e9, topk_ms)
        print(line, flush=True)
        c.close()


if __name__ == "__main__":
    main()
