"""GPU-box tool: ms per 1024-query batch through the int8 matrix-core filter (forced), 10M x 384 f32, METRICS=4,1,3 (dot, L2, cosine); VG_LIB_PATH selects a variant build (tools/build_q8_variants.sh)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VG_BATCH_Q8"] = "1"
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, nq, k = int(os.environ.get("ROWS", "10000000")), 384, 1024, 20
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda"); 
for b in range(n // 500000):
    gen.manual_seed(42 * 100003 + b)
    t = torch.randn((500000, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), 500000, dim * 4); del t
qs = np.random.default_rng(44).standard_normal((nq, dim), dtype=np.float32)
for metric in [int(m) for m in os.environ.get("METRICS", "4").split(",")]:
    for i in range(2): c.scan_topk_batch(metric, qs, k)
    t0 = time.perf_counter()
    for i in range(5): c.scan_topk_batch(metric, qs, k)
    print(os.environ.get("VG_LIB_PATH", "default").split("/")[-1], "metric", metric, "path", c.last_batch_path(), "ms/batch %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
