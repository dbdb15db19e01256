"""GPU-box tool (round 6, needs a -DVG_LAB library): the default single-query filter path against the size of its plain pre-pass
(VG_SCAN_FILTER_PREPASS_DIV: 1 / this of the rows) - wall ms per query (events off), exact evaluations per query."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, k = int(os.environ.get("ROWS", "10000000")), int(os.environ.get("DIM", "384")), 20
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda")
for b in range(n // 500000):
    gen.manual_seed(42 * 100003 + b)
    t = torch.randn((500000, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), 500000, dim * 4); del t
qs = np.random.default_rng(44).standard_normal((256, dim), dtype=np.float32)
for rnd in range(2):
    for div in [int(v) for v in os.environ.get("DIVS", "0,128,256,512,1024,2048").split(",")]:
        if div: os.environ["VG_SCAN_FILTER_PREPASS_DIV"] = str(div)
        else: os.environ.pop("VG_SCAN_FILTER_PREPASS_DIV", None)
        pkg.reload_switches()
        for q in qs[:16]: c.scan_topk(pkg.L2, q, k)
        c.filter_exact_evals()
        t0 = time.perf_counter()
        for q in qs: c.scan_topk(pkg.L2, q, k)
        ms = (time.perf_counter() - t0) / len(qs) * 1e3
        print("round", rnd, "prepass 1 /", div or "default", "ms/query %.4f" % ms, "evals/query %.0f" % (c.filter_exact_evals() / len(qs)), flush=True)
