#!/usr/bin/env python3
"""GPU-box tool: achieved HBM GB/s of the scan kernel for every (type, metric) at one row size.
Scratch measurement aid (not part of the product or of bench.py's contract).
    python tools_kernel_matrix.py [--rows 4000000] [--dim 384]
"""
import argparse
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dims", type=str, default="384,768")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--types", type=str, default="1,2,3,4,5")
    ap.add_argument("--bytes", type=float, default=0, help="if set: rows = bytes / row size (same HBM footprint for every dim)")
    ap.add_argument("--filter", type=int, default=-1, help="vg_corpus_set_scan_filter mode: 0 = plain scans, 1 = filter scans where served, -1 = default")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    names = {1: "f32", 2: "f16", 3: "bf16", 4: "u8", 5: "i8"}
    mnames = {1: "l2", 2: "sql2", 3: "cos", 4: "dot", 5: "l1"}
    for dim in [int(x) for x in args.dims.split(",")]:
        for vt in [int(x) for x in args.types.split(",")]:
            es = pkg.TYPE_SIZE[vt]
            n = args.rows if not args.bytes else int(args.bytes // (dim * es))
            c = pkg.Corpus(vt, dim, capacity=n)
            blk = 1_000_000
            for r0 in range(0, n, blk):
                nr = min(blk, n - r0)
                if vt == 1:
                    t = torch.randn((nr, dim), device="cuda", dtype=torch.float32)
                elif vt == 2:
                    t = torch.randn((nr, dim), device="cuda", dtype=torch.float16)
                elif vt == 3:
                    t = torch.randn((nr, dim), device="cuda", dtype=torch.bfloat16)
                elif vt == 4:
                    t = torch.randint(0, 256, (nr, dim), device="cuda", dtype=torch.uint8)
                else:
                    t = torch.randint(-128, 128, (nr, dim), device="cuda", dtype=torch.int8)
                torch.cuda.synchronize()
                c.append_device(t.data_ptr(), nr, dim * es)
                del t
            rng = np.random.default_rng(1)
            if vt == 1:
                q = rng.standard_normal(dim, dtype=np.float32)
            elif vt == 2:
                q = rng.standard_normal(dim, dtype=np.float32).astype(np.float16).view(np.uint16)
            elif vt == 3:
                q = (rng.standard_normal(dim, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
            elif vt == 4:
                q = rng.integers(0, 256, dim).astype(np.uint8)
            else:
                q = rng.integers(-128, 128, dim).astype(np.int8)
            c.set_profiling(True)
            c.set_scan_filter(args.filter)
            line = "%-5s dim %4d rows %d :" % (names[vt], dim, n)
            for m in (1, 3, 4, 5):
                for _ in range(2):
                    c.scan_topk(m, q, 20)
                c.set_profiling(True)
                c.filter_exact_evals()
                for _ in range(args.reps):
                    c.scan_topk(m, q, 20)
                nl, scan_ms, merge_ms, pre_ms = c.profile_mean_ms_ex()
                gbs = n * dim * es / (scan_ms * 1e-3) / 1e9
                line += "  %s %.3f ms %5.0f GB/s" % (mnames[m], scan_ms, gbs)
                if c.kernel_name(m).startswith("scan_filter"):
                    line += " (filter: pre-pass %.0f us, %d exact rows/query)" % (pre_ms * 1e3, c.filter_exact_evals() // args.reps)
            print(line + "   [" + ", ".join(c.kernel_name(m) for m in (1, 3, 4, 5)) + "]", flush=True)
            c.close()


if __name__ == "__main__":
    main()
