#!/usr/bin/env python3
"""GPU-box tool: what vg_corpus_patch_rows / vg_corpus_delete_rows cost on a 10M x 384 f32 corpus: the call
itself, and the first scan afterwards (which re-makes the derived per-row data from the first touched row on).
    python tools/tools_row_maintenance.py [--rows 10000000] [--dim 384]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    n, dim = args.rows, args.dim
    c = pkg.Corpus(pkg.F32, dim, capacity=n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for r0 in range(0, n, 1_000_000):
        t = torch.randn((min(1_000_000, n - r0), dim), generator=gen, device="cuda")
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), t.shape[0], dim * 4)
        del t
    rng = np.random.default_rng(2)
    q = rng.standard_normal(dim, dtype=np.float32)

    def timed(f):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r

    print("first scan (builds the int8 shadow copy + norms): %.2f ms" % timed(lambda: c.scan_topk(pkg.L2, q, 20))[0])
    print("steady scan: %.3f ms" % timed(lambda: c.scan_topk(pkg.L2, q, 20))[0])
    for npatch, where in ((1, "last row"), (100, "random rows"), (1, "row 0")):
        pos = {"last row": np.array([n - 1]), "row 0": np.array([0])}.get(where, rng.permutation(n)[:npatch]).astype(np.int64)
        new = rng.standard_normal((len(pos), dim), dtype=np.float32)
        ms, _ = timed(lambda: c.patch_rows(pos, new))
        ms2, _ = timed(lambda: c.scan_topk(pkg.L2, q, 20))
        print("patch %4d %-12s: call %.3f ms, next scan %.2f ms (derived data re-made from row %d on)" % (len(pos), where, ms, ms2, int(pos.min())))
        print("   steady scan again: %.3f ms" % timed(lambda: c.scan_topk(pkg.L2, q, 20))[0])
    for ndel, where in ((1, "last row"), (100, "random rows"), (1, "row 0"), (4096, "random rows")):
        m = c.rows
        pos = {"last row": np.array([m - 1]), "row 0": np.array([0])}.get(where, np.sort(rng.permutation(m)[:ndel])).astype(np.int64)
        ms, _ = timed(lambda: c.delete_rows(pos))
        ms2, _ = timed(lambda: c.scan_topk(pkg.L2, q, 20))
        print("delete %4d %-12s: call %.2f ms (rows behind move up on the device), next scan %.2f ms" % (len(pos), where, ms, ms2))
    print("for comparison - re-staging %d rows from host memory (what a re-stage costs WITHOUT sqlite3_step): " % 1_000_000, end="")
    host = rng.standard_normal((1_000_000, dim), dtype=np.float32)
    c2 = pkg.Corpus(pkg.F32, dim, capacity=1_000_000)
    ms, _ = timed(lambda: c2.append(host))
    print("%.1f ms per million rows" % ms)


if __name__ == "__main__":
    main()
