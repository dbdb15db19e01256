#!/usr/bin/env python3
"""Where a wavefront of the int8 batch kernel spends its time (needs a -DVGI_TIMING=1 build of vg_batch_i8.hip:
tools/build_i8_variants.sh timing -DVGI_TIMING=1; VG_LIB_PATH=.../libvectorgpu_timing.so).
    python tools/tools_i8_timing.py [--rows 10000000] [--dim 768] [--nq 1024] [--metric 3]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--metric", type=int, default=3)
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    lib = C.CDLL(pkg.LIB_PATH)
    c = pkg.Corpus(pkg.U8, args.dim, capacity=args.rows)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    for r0 in range(0, args.rows, 1_000_000):
        nr = min(1_000_000, args.rows - r0)
        t = torch.randint(0, 256, (nr, args.dim), generator=gen, device="cuda", dtype=torch.uint8)
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), nr, args.dim)
        del t
    rng = np.random.default_rng(44)
    qs = rng.integers(0, 256, (args.nq, args.dim)).astype(np.uint8)
    c.scan_topk_batch(args.metric, qs, 20)
    out = (C.c_ulonglong * 16)()
    lib.vg_batch_i8_timing(out, 1)
    c.set_profiling(True)
    c.scan_topk_batch(args.metric, qs, 20)
    lib.vg_batch_i8_timing(out, 0)
    v = [int(x) for x in out]
    tiles = max(v[6], 1)
    print("dim %d metric %d nq %d: kernel ms (events) %.3f  wave-tiles %d  ticks per wave-tile (whole loop) %.1f" %
          (args.dim, args.metric, args.nq, c.profile_mean_ms()[1], v[6], v[0] / tiles))
    for n, x in zip(["MFMA role (k loop of a double tile, two chains)", "X role: DMA issue", "X role: tests + slow paths (+ DMA wait)",
                     "barrier"], [v[3], v[4], v[2], v[1]]):
        print("  %-58s %5.1f %%   %8.1f ticks per wave-tile" % (n, 100.0 * x / max(v[0], 1), x / tiles))
    print("  double tiles with a slow path: %.2f %% of the wave-tiles" % (100.0 * v[7] / tiles))
    print("  barrier wait per wavefront index 0..7 (ticks per tile): " + " ".join("%.0f" % (8.0 * x / tiles) for x in v[8:16]))
    c.close()


if __name__ == "__main__":
    main()
