#!/usr/bin/env python3
"""GPU-box measurement of BASELINE config C5: nq queries x N x 384 f32, dot product, top-20, batched on the matrix
cores (vg_scan_topk_batch -> vg_batch_kernel).  Prints one JSON line per nq with the MFMA roofline
(flops = 2 * nq * N * D per launch, peak = 157.3 TF f32 MFMA, MI355X_MICROARCH.md).
    python tools_batch_bench.py [--rows 10000000] [--nq 1024,256,128] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F32_MFMA_PEAK_TF = 157.3
PEAK_TF = {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0, "u8": 3944.0, "i8": 3944.0}   # dense MFMA peaks (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--nq", type=str, default="1024,256,128")
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--metric", type=int, default=4)
    ap.add_argument("--type", type=str, default="f32", choices=("f32", "u8", "i8", "f16", "bf16"))
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    n, dim = args.rows, args.dim
    vt = {"f32": pkg.F32, "u8": pkg.U8, "i8": pkg.I8, "f16": pkg.F16, "bf16": pkg.BF16}[args.type]
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}.get(args.type)
    es = pkg.TYPE_SIZE[vt]
    c = pkg.Corpus(vt, dim, capacity=n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    for r0 in range(0, n, 1_000_000):
        nr = min(1_000_000, n - r0)
        if tdt is not None:
            t = torch.randn((nr, dim), generator=gen, device="cuda", dtype=torch.float32).to(tdt)
        elif vt == pkg.U8:
            t = torch.randint(0, 256, (nr, dim), generator=gen, device="cuda", dtype=torch.uint8)
        else:
            t = torch.randint(-128, 128, (nr, dim), generator=gen, device="cuda", dtype=torch.int8)
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), nr, dim * es)
        del t
    c.set_profiling(True)
    rng = np.random.default_rng(44)
    for nq in [int(x) for x in args.nq.split(",")]:
        if vt == pkg.F32:
            qs = rng.standard_normal((nq, dim), dtype=np.float32)
        elif tdt is not None:
            qs = torch.from_numpy(rng.standard_normal((nq, dim), dtype=np.float32)).to(tdt).view(torch.int16).numpy().view(np.uint16)
        elif vt == pkg.U8:
            qs = rng.integers(0, 256, (nq, dim)).astype(np.uint8)
        else:
            qs = rng.integers(-128, 128, (nq, dim)).astype(np.int8)
        c.scan_topk_batch(args.metric, qs, args.k)                  # warm-up
        c.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ids, dist, cnt = c.scan_topk_batch(args.metric, qs, args.k)
        wall = (time.perf_counter() - t0) / args.reps
        nl, kern_ms, _ = c.profile_mean_ms()
        flops = 2.0 * nq * n * dim
        tf = flops / (kern_ms * 1e-3) / 1e12
        # spot check of query 0 against a single-query scan (exact reference arithmetic)
        one_ids, one_dist = c.scan_topk(args.metric, qs[0], args.k)
        agree = float(np.mean(np.isin(ids[0], one_ids)))
        print(json.dumps({
            "workload": "%d queries x %dx%d %s %s top-%d, batched MFMA" % (nq, n, dim, args.type, {1: "L2", 2: "squared L2", 3: "cosine", 4: "dot", 5: "L1"}[args.metric], args.k),
            "kernel_ms": kern_ms, "wall_ms_per_batch": wall * 1e3, "queries_per_s": nq / wall,
            "query_vector_pairs_per_s": nq * n / wall,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TF[args.type], "unit": "TFLOP/s", "frac": tf / PEAK_TF[args.type],
                         "flops_per_launch": flops},
            "speedup_vs_single_query_scans": (nq * 2.28e-3) / wall,
            "top20_overlap_with_single_query_path_q0": agree}), flush=True)
    c.close()


if __name__ == "__main__":
    main()
