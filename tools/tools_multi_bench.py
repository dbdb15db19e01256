#!/usr/bin/env python3
"""GPU-box measurement of the multi-query scan (vg_scan_multi_kernel), the batch path of the shapes the matrix-core
kernels do not serve: wall time per batch with VG_MULTI_SCAN=1 against nq single scans (VG_MULTI_SCAN=0), plus a
result comparison of the two.
    python tools_multi_bench.py [--rows 10000000] [--nq 64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # type, dim, metric, k
    ("f32", 384, 5, 20), ("f32", 768, 1, 20), ("f32", 1536, 4, 20), ("f32", 384, 1, 50),
    ("u8", 768, 5, 20), ("i8", 1536, 1, 20), ("u8", 128, 5, 20),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--nq", type=int, default=64)
    ap.add_argument("--bytes", type=float, default=16e9)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    names = {1: "L2", 2: "squared L2", 3: "cosine", 4: "dot", 5: "L1"}
    for tname, dim, metric, k in CASES:
        vt = {"f32": pkg.F32, "f16": pkg.F16, "bf16": pkg.BF16, "u8": pkg.U8, "i8": pkg.I8}[tname]
        es = pkg.TYPE_SIZE[vt]
        n = int(min(args.rows, args.bytes // (dim * es)))
        c = pkg.Corpus(vt, dim, capacity=n)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(42)
        for r0 in range(0, n, 1_000_000):
            nr = min(1_000_000, n - r0)
            if tname == "u8":
                t = torch.randint(0, 256, (nr, dim), generator=gen, device="cuda", dtype=torch.uint8)
            elif tname == "i8":
                t = torch.randint(-128, 128, (nr, dim), generator=gen, device="cuda", dtype=torch.int8)
            else:
                t = torch.randn((nr, dim), generator=gen, device="cuda", dtype=torch.float32)
                t = t.to({"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[tname])
            torch.cuda.synchronize()
            c.append_device(t.data_ptr(), nr, dim * es)
            del t
        rng = np.random.default_rng(44)
        if tname == "u8":
            qs = rng.integers(0, 256, (args.nq, dim)).astype(np.uint8)
        elif tname == "i8":
            qs = rng.integers(-128, 128, (args.nq, dim)).astype(np.int8)
        else:
            qf = torch.from_numpy(rng.standard_normal((args.nq, dim), dtype=np.float32))
            qt = qf.to({"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[tname])
            qs = qt.view(torch.int16).numpy() if tname in ("f16", "bf16") else qt.numpy()
        res = {}
        for flag in ("1", "0"):
            os.environ["VG_MULTI_SCAN"] = flag
            c.scan_topk_batch(metric, qs, k)
            t0 = time.perf_counter()
            for _ in range(3):
                out = c.scan_topk_batch(metric, qs, k)
            res[flag] = ((time.perf_counter() - t0) / 3, out)
        (w1, (i1, d1, c1)), (w0, (i0, d0, c0)) = res["1"], res["0"]
        same_ids = bool(np.array_equal(i1, i0))
        maxrel = float(np.max(np.abs(d1 - d0) / np.maximum(np.abs(d0), 1e-30)))
        bytes_per_scan = n * dim * es
        print(json.dumps({"case": "%d q x %dx%d %s %s top-%d" % (args.nq, n, dim, tname, names[metric], k),
                          "multi_ms": w1 * 1e3, "singles_ms": w0 * 1e3, "speedup": w0 / w1,
                          "multi_effective_TBps": args.nq * bytes_per_scan / w1 / 1e12,
                          "single_TBps": args.nq * bytes_per_scan / w0 / 1e12,
                          "same_ids": same_ids, "max_rel_dist_diff": maxrel}), flush=True)
        c.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
