#!/usr/bin/env python3
"""GPU-box tool: one line per libvectorgpu build (VG_LIB_PATH) - what a 1024-query batch over a 10M x 768 uint8 corpus costs.  Run over the
ablation builds of vg_batch_i8.hip (tools/build_i8_variants.sh abN -DVGI_ABLATE=N: wrong results, timing only) it gives the ladder of
DESIGN.md 3c: full kernel | no slow path | no fast test | no DMA | no barrier (MFMAs + B-operand reads only).
    VG_LIB_PATH=.../libvectorgpu_i8_ab1.so python tools/tools_i8_ablate.py [--rows 10000000] [--dim 768] [--nq 1024] [--metric 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--metric", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    c = pkg.Corpus(pkg.U8, args.dim, capacity=args.rows)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    for r0 in range(0, args.rows, 1_000_000):
        nr = min(1_000_000, args.rows - r0)
        t = torch.randint(0, 256, (nr, args.dim), generator=gen, device="cuda", dtype=torch.uint8)
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), nr, args.dim)
        del t
    qs = np.random.default_rng(44).integers(0, 256, (args.nq, args.dim)).astype(np.uint8)
    c.scan_topk_batch(args.metric, qs, 20)
    c.scan_topk_batch(args.metric, qs, 20)
    lat = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        c.scan_topk_batch(args.metric, qs, 20)
        lat.append(time.perf_counter() - t0)
    ms = float(np.median(lat)) * 1e3
    ops = 2.0 * args.nq * args.rows * ((args.dim + 31) // 32 * 32)
    print(json.dumps({"lib": os.path.basename(pkg.LIB_PATH), "rows": args.rows, "dim": args.dim, "nq": args.nq, "metric": args.metric,
                      "ms_per_batch": round(ms, 3), "TOPs": round(ops / (ms * 1e-3) / 1e12, 1), "frac_of_3944": round(ops / (ms * 1e-3) / 1e12 / 3944, 4),
                      "path": c.last_batch_path()}))
    c.close()


if __name__ == "__main__":
    main()
