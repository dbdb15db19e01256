"""GPU-box tool: small batches (NQ queries) over 10M x 384 f32: the default policy vs the int8 filter forced"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, k = 10_000_000, 384, 20
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda")
for b in range(n // 500000):
    gen.manual_seed(42 * 100003 + b)
    t = torch.randn((500000, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), 500000, dim * 4); del t
for nq in [int(x) for x in os.environ.get("NQS", "16,64,128,256,512").split(",")]:
    qs = np.random.default_rng(44).standard_normal((nq, dim), dtype=np.float32)
    for force in (None, "1"):
        if force: os.environ["VG_BATCH_Q8"] = force
        else: os.environ.pop("VG_BATCH_Q8", None)
        pkg.reload_switches()
        for i in range(2): r0 = c.scan_topk_batch(4, qs, k)
        t0 = time.perf_counter()
        for i in range(4): r = c.scan_topk_batch(4, qs, k)
        print("nq", nq, "VG_BATCH_Q8", force, "path", c.last_batch_path(), "ms/batch %.3f" % ((time.perf_counter() - t0) / 4 * 1e3), flush=True)
