#!/usr/bin/env python3
"""GPU-box tool: how selective the filter scans' lower bound is on NON-Gaussian data: unit-norm embeddings
clustered around centres (|q||x| = 1, so the bound's slack is large against the distance gaps), and a corpus of heavy
near-duplicates.  Prints, per corpus type and metric, the filter kernel's time, the rows it evaluated exactly per query
(vg_filter_exact_evals) and the plain kernel's time next to it.
    python tools/tools_filter_selectivity.py [--rows 10000000] [--dim 384]"""
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
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--types", default="f32,f16,bf16")
    ap.add_argument("--data", default="clustered,near,gaussian")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    n, dim = args.rows, args.dim
    mnames = {1: "l2", 3: "cos", 4: "dot", 5: "l1"}
    shadow = os.environ.get("VG_SCAN_FILTER_SHADOW", "int8 (default)")
    for data in ("clustered unit-norm (1000 centres, noise 0.3)", "near-duplicates (20000 distinct rows + 1e-3 noise)", "gaussian N(0,1)"):
        if data.split()[0].split("-")[0] not in args.data.split(","):
            continue
        for tname, vt, tdt in (("f32", pkg.F32, torch.float32), ("f16", pkg.F16, torch.float16), ("bf16", pkg.BF16, torch.bfloat16),
                               ("u8", pkg.U8, torch.uint8), ("i8", pkg.I8, torch.int8)):
            if tname not in args.types.split(","):
                continue
            gen = torch.Generator(device="cuda")
            gen.manual_seed(7)
            ncent = 1000 if data.startswith("clustered") else 20000
            noise = 0.3 if data.startswith("clustered") else 1e-3
            cent = torch.randn((ncent, dim), generator=gen, device="cuda")
            cent /= cent.norm(dim=1, keepdim=True)
            c = pkg.Corpus(vt, dim, capacity=n)
            es = pkg.TYPE_SIZE[vt]
            keep = None
            for r0 in range(0, n, 1_000_000):
                nr = min(1_000_000, n - r0)
                idx = torch.randint(0, ncent, (nr,), generator=gen, device="cuda")
                if data.startswith("gaussian"):
                    x = torch.randn((nr, dim), generator=gen, device="cuda")
                else:
                    x = cent[idx] + noise / (dim ** 0.5) * torch.randn((nr, dim), generator=gen, device="cuda")
                    x /= x.norm(dim=1, keepdim=True)
                if vt in (pkg.U8, pkg.I8):      # quantized like vector_quantize would: one scale for the table (+- 4.5 sigma of an element)
                    sig = 1.0 if data.startswith("gaussian") else 1.0 / dim ** 0.5
                    if vt == pkg.U8:
                        x = torch.clamp(torch.round((x + 4.5 * sig) * (255.0 / (9.0 * sig))), 0, 255)
                    else:
                        x = torch.clamp(torch.round(x * (127.0 / (4.5 * sig))), -128, 127)
                t = x.to(tdt).contiguous()
                torch.cuda.synchronize()
                c.append_device(t.data_ptr(), nr, dim * es)
                if keep is None:
                    keep = t[:64].clone()
                del t, x
            qs = keep.view(torch.uint8).cpu().numpy().view({4: np.float32, 2: np.uint16, 1: np.uint8}[es]).reshape(64, dim)
            line = "%-52s %-4s %dx%d%s:" % (data, tname, n, dim, " [shadow " + shadow + "]")
            for m in (1, 3, 4) + ((5,) if vt in (pkg.F16, pkg.BF16) else ()):
                res = {}
                for mode in (1, 0):
                    c.set_scan_filter(mode)
                    c.scan_topk(m, qs[0], 20)
                    c.set_profiling(True)
                    c.filter_exact_evals()
                    for i in range(args.reps):
                        c.scan_topk(m, qs[1 + i], 20)
                    nl, scan_ms, merge_ms, pre_ms = c.profile_mean_ms_ex()
                    res[mode] = (scan_ms, pre_ms, c.filter_exact_evals() // args.reps)
                line += "  %s filter %.3f ms (+%.0f us pre-pass, %d exact rows/query) plain %.3f ms;" % (
                    mnames[m], res[1][0], res[1][1] * 1e3, res[1][2], res[0][0])
            print(line, flush=True)
            c.close()


if __name__ == "__main__":
    main()
