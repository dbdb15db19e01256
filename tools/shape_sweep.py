#!/usr/bin/env python3
"""GPU-box tool: every exact-cover launch shape (lanes per row x chunks per lane) of the plain scan kernels, per type / dim / metric,
through the VG_LPR_LOG2 / VG_U experiment overrides.  One line per (type, dim, shape).  Scratch measurement aid.
    python tools/shape_sweep.py [--rows 10000000] [--cases 2:384,3:384,4:384,4:768,2:768,3:768,1:384]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--bytes", type=float, default=7.68e9)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cases", type=str, default="2:384,3:384,4:384,4:768,2:768,3:768,1:384")
    ap.add_argument("--repeat", type=int, default=1, help="walk the shape list this many times (A/B/A/B against drift)")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    names = {1: "f32", 2: "f16", 3: "bf16", 4: "u8", 5: "i8"}
    mnames = {1: "l2", 3: "cos", 4: "dot", 5: "l1"}
    for case in args.cases.split(","):
        vt, dim = [int(x) for x in case.split(":")]
        es = pkg.TYPE_SIZE[vt]
        n = min(args.rows, int(args.bytes // (dim * es)))
        c = pkg.Corpus(vt, dim, capacity=n)
        for r0 in range(0, n, 1_000_000):
            nr = min(1_000_000, n - r0)
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
        q32 = rng.standard_normal(dim, dtype=np.float32)
        q = {1: q32, 2: q32.astype(np.float16).view(np.uint16), 3: (q32.view(np.uint32) >> 16).astype(np.uint16),
             4: rng.integers(0, 256, dim).astype(np.uint8), 5: rng.integers(-128, 128, dim).astype(np.int8)}[vt]
        c.set_scan_filter(0)
        nch = (dim * es + 15) // 16
        shapes = [(-1, -1)]
        for l2 in range(0, 7):
            for U in (1, 2, 3, 4, 6, 8):
                if (1 << l2) * U == nch or ((1 << l2) * U > nch and (1 << l2) * U * 3 <= nch * 4 and U <= 6 and l2 >= 2):
                    shapes.append((l2, U))
        for l2, U in shapes * args.repeat:
            if l2 < 0:
                os.environ.pop("VG_LPR_LOG2", None); os.environ.pop("VG_U", None)
            else:
                os.environ["VG_LPR_LOG2"] = str(l2); os.environ["VG_U"] = str(U)
            line = "%-5s dim %4d rows %d shape %-12s:" % (names[vt], dim, n, "default" if l2 < 0 else "lpr%d_u%d" % (1 << l2, U))
            for m in (1, 3, 4, 5):
                for _ in range(2):
                    c.scan_topk(m, q, 20)
                c.set_profiling(True)
                for _ in range(args.reps):
                    c.scan_topk(m, q, 20)
                nl, scan_ms, merge_ms, pre_ms = c.profile_mean_ms_ex()
                line += "  %s %5.0f" % (mnames[m], n * dim * es / (scan_ms * 1e-3) / 1e9)
            print(line + "  GB/s  [" + c.kernel_name(1) + "]", flush=True)
        os.environ.pop("VG_LPR_LOG2", None); os.environ.pop("VG_U", None)
        c.close()


if __name__ == "__main__":
    main()
