#!/usr/bin/env python3
"""Where a wavefront of the half-precision batch kernel spends its time (needs a -DVGH_TIMING=1 build of
vg_batch_h.hip: tools/build_half_variants.sh timing -DVGH_TIMING=1; VG_LIB_PATH=.../libvectorgpu_timing.so).
    python tools/tools_half_timing.py [--rows 10000000] [--dim 384] [--nq 1024] [--type f16] [--metric 4]"""
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
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--metric", type=int, default=4)
    ap.add_argument("--type", default="f16", choices=("f16", "bf16", "f32"), help="f32: an f32 corpus through the bf16 filter (VG_F32_FILTER=1)")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    import __graft_entry__ as g
    pkg = g.load_package()
    lib = pkg.lib()
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.type]
    vt = {"f16": pkg.F16, "bf16": pkg.BF16, "f32": pkg.F32}[args.type]
    if args.type == "f32":
        os.environ["VG_F32_FILTER"] = "1"
    c = pkg.Corpus(vt, args.dim, capacity=args.rows)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    for r0 in range(0, args.rows, 1_000_000):
        nr = min(1_000_000, args.rows - r0)
        t = torch.randn((nr, args.dim), generator=gen, device="cuda", dtype=torch.float32).to(tdt)
        torch.cuda.synchronize()
        c.append_device(t.data_ptr(), nr, args.dim * pkg.TYPE_SIZE[vt])
        del t
    rng = np.random.default_rng(44)
    qs = rng.standard_normal((args.nq, args.dim), dtype=np.float32)
    if args.type != "f32":
        qs = torch.from_numpy(qs).to(tdt).view(torch.int16).numpy().view(np.uint16)
    c.scan_topk_batch(args.metric, qs, 20)
    out = (C.c_ulonglong * 16)()
    lib.vg_batch_h_timing(out, 1)
    c.set_profiling(True)
    c.scan_topk_batch(args.metric, qs, 20)
    lib.vg_batch_h_timing(out, 0)
    v = [int(x) for x in out]
    names = ["k loop", "filter", "survivors", "DMA wait", "barrier"]
    tot = sum(v[:5])
    print("kernel ms (events):", c.profile_mean_ms()[1], " wave-tiles:", v[6], " ticks per wave-tile:", tot / max(v[6], 1))
    for n, x in zip(names, v[:5]):
        print("  %-10s %5.1f %%   %8.1f ticks per wave-tile" % (n, 100.0 * x / tot, x / max(v[6], 1)))
    print("  whole kernel vs sum of phases: %.3f" % (v[5] / tot))
    print("  wave-tiles with survivors: %.2f %%   exact evaluations: %d (%.2f per query and partition-list of the run)" %
          (100.0 * (v[7] & 0xFFFFFFFF) / max(v[6], 1), v[7] >> 32, (v[7] >> 32) / max(args.nq, 1)))
    ne = max(v[9], 1)
    print("  real pass: %d survivor-phase entries (%.2f %% of wave-tiles), %.2f pending registers and %.2f evaluated pairs per entry;"
          % (v[9], 100.0 * v[9] / max(v[6], 1), v[8] / ne, v[13] / ne))
    print("             %.0f ticks per entry, of which exact evaluation (loads + f64 sums) %.0f per pair" % (v[12] / ne, v[10] / max(v[13], 1)))
    c.close()


if __name__ == "__main__":
    main()
