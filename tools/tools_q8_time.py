"""GPU-box tool: ms per 1024-query batch through the int8 matrix-core filter (forced), ROWS x DIM f32 (default 10M x 384), METRICS=4,1,3 (dot, L2,
cosine) + exact evaluations per query; VG_LIB_PATH selects a variant build (tools/build_q8_variants.sh)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VG_BATCH_Q8"] = "1"
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, nq, k = int(os.environ.get("ROWS", "10000000")), int(os.environ.get("DIM", "384")), int(os.environ.get("NQ", "1024")), int(os.environ.get("K", "20"))
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda")
blk = 500000 if dim <= 512 else 125000
for b in range(n // blk):
    gen.manual_seed(42 * 100003 + b)
    t = torch.randn((blk, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), blk, dim * 4); del t
qs = np.random.default_rng(44).standard_normal((nq, dim), dtype=np.float32)
reps = int(os.environ.get("REPS", "5"))
for metric in [int(m) for m in os.environ.get("METRICS", "4").split(",")]:
    for i in range(2): c.scan_topk_batch(metric, qs, k)
    c.batch_filter_exact_evals()
    t0 = time.perf_counter()
    for i in range(reps): c.scan_topk_batch(metric, qs, k)
    ms = (time.perf_counter() - t0) / reps * 1e3
    ev = c.batch_filter_exact_evals() / float(reps * nq)
    print(os.environ.get("VG_LIB_PATH", "default").split("/")[-1], "dim", dim, "metric", metric, "path", c.last_batch_path(), "ms/batch %.3f" % ms, "evals/query %.0f" % ev, flush=True)
