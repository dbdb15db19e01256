#!/usr/bin/env python3
"""GPU-box tool: end-to-end vg_scan_topk latency vs k on a 10M x 384 f32 corpus (k <= 64 fused lists, k > 64 ...)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import __graft_entry__ as g
pkg = g.load_package()
n, dim = int(os.environ.get("ROWS", 10_000_000)), 384
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda"); gen.manual_seed(42)
for r0 in range(0, n, 1_000_000):
    nr = min(1_000_000, n - r0)
    t = torch.randn((nr, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), nr, dim * 4); del t
qs = np.random.default_rng(1).standard_normal((8, dim), dtype=np.float32)
for k in (20, 64, 65, 100, 256, 1000, 10000):
    c.scan_topk(pkg.L2, qs[0], k)
    t0 = time.perf_counter()
    for i in range(8):
        ids, dist = c.scan_topk(pkg.L2, qs[i], k)
    dt = (time.perf_counter() - t0) / 8
    print("k=%5d: %.3f ms per query (returned %d, sorted %s)" % (k, dt * 1e3, len(ids), bool(np.all(np.diff(dist) >= 0))), flush=True)
c.set_profiling(True)
for k in (20, 100):
    for i in range(8):
        c.scan_topk(pkg.L2, qs[i], k)
    print("k=%d scan kernel (events): %.3f ms" % (k, c.last_kernel_ms()[0]))
