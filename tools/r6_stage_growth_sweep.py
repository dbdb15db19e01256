"""GPU-box tool (round 6): late-stage growth of the int8 batch filter's schedule (VG_BATCH_STAGES = percent, > 100) - ms per batch and exact
evaluations per query, one corpus, DIM / ROWS / METRICS from the environment."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VG_BATCH_Q8"] = "1"
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, nq, k = int(os.environ.get("ROWS", "10000000")), int(os.environ.get("DIM", "1536")), int(os.environ.get("NQ", "1024")), 20
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda")
blk = 500000 if dim <= 512 else 125000
for b in range(n // blk):
    gen.manual_seed(42 * 100003 + b)
    t = torch.randn((blk, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), blk, dim * 4); del t
qs = np.random.default_rng(44).standard_normal((nq, dim), dtype=np.float32)
metric = int(os.environ.get("METRICS", "4"))
for rnd in range(2):
    for growth in [int(v) for v in os.environ.get("GROWTHS", "0,125,150,175,200,300").split(",")]:
        if growth: os.environ["VG_BATCH_STAGES"] = str(growth)
        else: os.environ.pop("VG_BATCH_STAGES", None)
        pkg.reload_switches()
        for i in range(2): c.scan_topk_batch(metric, qs, k)
        c.batch_filter_exact_evals()
        t0 = time.perf_counter()
        for i in range(4): c.scan_topk_batch(metric, qs, k)
        ms = (time.perf_counter() - t0) / 4 * 1e3
        print("round", rnd, "dim", dim, "metric", metric, "growth", growth or "default", "ms/batch %.3f" % ms, "evals/query %.0f" % (c.batch_filter_exact_evals() / (4.0 * nq)), flush=True)
