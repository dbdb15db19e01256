#!/usr/bin/env python3
"""C5 through the product's default path (bf16 matrix-core filter over an f32 corpus): sweep the two launch knobs of the staged
passes - VG_BATCH_PREPASS (the bound-only pre-pass covers 1/n of the rows) x VG_BATCH_STAGES (growth of the staged real passes in
percent, 0 = one pass) - in ONE process over one resident corpus (the library reads both per launch).
    python tools/batch_knob_sweep.py [--rows 10000000] [--nq 1024] [--type f32|f16|bf16]"""
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
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--type", default="f32")
    ap.add_argument("--metric", type=int, default=4)
    ap.add_argument("--no-filter", action="store_true", help="f32: the f32 MFMA kernel (vg_batch.hip) instead of the bf16 filter")
    ap.add_argument("--pre", default="32,16,64,128,256")
    ap.add_argument("--stages", default="400,0,200,800,1600")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    import bench
    pkg = g.load_package()
    vt = {"f32": pkg.F32, "f16": pkg.F16, "bf16": pkg.BF16, "u8": pkg.U8, "i8": pkg.I8}[args.type]
    dim = 768 if vt in (pkg.U8, pkg.I8) else 384
    c = bench.make_shard(pkg, torch, vt, dim, args.rows, 42, 0)
    if vt == pkg.F32:
        os.environ["VG_F32_FILTER"] = "0" if args.no_filter else "1"
    rng = np.random.default_rng(44)
    qf = rng.standard_normal((args.nq, dim), dtype=np.float32)
    if vt == pkg.U8:
        q = bench.c3_queries(args.nq, dim)
    elif vt == pkg.I8:
        q = np.clip(np.rint(qf * 40.0), -128, 127).astype(np.int8)
    else:
        q = qf if vt == pkg.F32 else (qf.astype(np.float16) if vt == pkg.F16 else torch.from_numpy(qf).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))
    base = None
    for pre in args.pre.split(","):
        for stages in args.stages.split(","):
            os.environ["VG_BATCH_PREPASS"] = pre
            os.environ["VG_BATCH_STAGES"] = stages
            ids, dist, cnt = c.scan_topk_batch(args.metric, q, 20)
            c.set_profiling(True)
            for _ in range(3):
                c.scan_topk_batch(args.metric, q, 20)
            n, ms, _ = c.profile_mean_ms()
            if base is None:
                base = ids.copy()
            print(json.dumps({"type": args.type, "metric": args.metric, "nq": args.nq, "prepass_1_over": int(pre), "stages_growth_pct": int(stages), "kernel_ms": round(ms, 3),
                              "same_rowids_as_default": bool(np.array_equal(ids, base))}), flush=True)
    c.close()


if __name__ == "__main__":
    main()
