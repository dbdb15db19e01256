"""GPU-box tool: 1024 x 10M x 768 uint8 cosine batches (the exact int8 matrix-core kernel): ms per batch; run under rocprofv3 --kernel-trace for the timeline"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, nq, k = int(os.environ.get("ROWS", "10000000")), 768, 1024, 20
metric = int(os.environ.get("METRICS", "3"))
c = pkg.Corpus(pkg.U8, dim, capacity=n)
gen = torch.Generator(device="cuda")
for b in range(n // 500000):
    gen.manual_seed(42 * 100003 + b)
    t = (torch.rand((500000, dim), generator=gen, device="cuda") * 255.0 + 0.5).floor().clamp(0, 255).to(torch.uint8); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), 500000, dim); del t
qs = np.clip(np.floor(np.random.default_rng(44).random((nq, dim), dtype=np.float32) * 255 + 0.5), 0, 255).astype(np.uint8)
for i in range(2): c.scan_topk_batch(metric, qs, k)
t0 = time.perf_counter()
for i in range(4): c.scan_topk_batch(metric, qs, k)
print("c3b metric", metric, "path", c.last_batch_path(), "ms/batch %.3f" % ((time.perf_counter() - t0) / 4 * 1e3), flush=True)
