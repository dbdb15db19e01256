#!/usr/bin/env python3
"""Staged real passes of the batched matrix-core kernels (vg_batch_common.h: VG_BATCH_STAGES = growth in percent, 0 = one
real pass): kernel time per setting on one resident corpus per type, and a check that every setting returns the same lists.
    python tools/r2k_stage_sweep.py [--rows 10000000] [--types u8,f16] [--stages 0,400,200,150]
"""
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
    ap.add_argument("--types", type=str, default="u8,f16")
    ap.add_argument("--stages", type=str, default="0,400,200,150")
    ap.add_argument("--nq", type=str, default="1024,256")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    n = args.rows
    shapes = {"u8": (768, (4, 3, 1)), "i8": (768, (4,)), "f16": (384, (4, 3, 1)), "bf16": (384, (4,)), "f32": (384, (4,)),
              "u8s": (128, (3,))}
    for tname in args.types.split(","):
        dim, metrics = shapes[tname]
        base = tname.rstrip("s")
        vt = {"f32": pkg.F32, "u8": pkg.U8, "i8": pkg.I8, "f16": pkg.F16, "bf16": pkg.BF16}[base]
        tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}.get(base)
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
            for metric in metrics:
                ref = None
                for st in args.stages.split(","):
                    os.environ["VG_BATCH_STAGES"] = st
                    c.scan_topk_batch(metric, qs, 20)                       # warm-up
                    c.set_profiling(True)
                    for _ in range(args.reps):
                        ids, dist, cnt = c.scan_topk_batch(metric, qs, 20)
                    _, kern_ms, _ = c.profile_mean_ms()
                    c.set_profiling(False)
                    same = None
                    if ref is None:
                        ref = (ids.copy(), dist.copy())
                    else:
                        same = bool(np.array_equal(ref[0], ids) and np.array_equal(ref[1].view(np.uint64), dist.view(np.uint64)))
                    print(json.dumps({"type": tname, "dim": dim, "nq": nq, "metric": metric, "stages": st,
                                      "kernel_ms": round(kern_ms, 4), "same_lists_as_first_setting": same}), flush=True)
        c.close()
    os.environ.pop("VG_BATCH_STAGES", None)


if __name__ == "__main__":
    main()
