"""GPU-box check: batch (int8 filter, forced) vs plain single scans on clustered unit-norm data: are the distance BITS the same?"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["VG_BATCH_Q8"] = "1"
import torch
import __graft_entry__ as g
import datagen as dg
pkg = g.load_package()
n, dim, k, nq = int(os.environ.get("ROWS", "2000000")), 384, 20, 512
centres = dg.clustered_centres(torch, 42, dim)
qs = dg.clustered_queries(torch, centres, 42, nq)
c = pkg.Corpus(pkg.F32, dim, capacity=n)
for b in range(n // 500000):
    t = dg.clustered_block(torch, centres, 42, b, 500000); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), 500000, dim * 4); del t
for m in (3, 4, 1):
    c.set_scan_filter(0)
    plain = [c.scan_topk(m, qs[i], k) for i in range(16)]
    c.set_scan_filter(-1)
    ids, dist, cnt = c.scan_topk_batch(m, qs, k)
    bad_ids = sum(int(not np.array_equal(plain[i][0], ids[i])) for i in range(16))
    bad_d = sum(int(not np.array_equal(plain[i][1], dist[i])) for i in range(16))
    worst = max(float(np.max(np.abs(np.asarray(plain[i][1], dtype=np.float64) - dist[i]) / np.maximum(np.abs(plain[i][1]), 1e-30))) for i in range(16))
    print(os.environ.get("VG_LIB_PATH", "default").split("/")[-1], "metric", m, "path", c.last_batch_path(), "queries with other rowids", bad_ids, "with other distance bits", bad_d, "worst rel %.3g" % worst)
    if bad_d:
        i = [i for i in range(16) if not np.array_equal(plain[i][1], dist[i])][0]
        print("   q", i, "plain", np.asarray(plain[i][1][:5]).view(np.uint32) if hasattr(plain[i][1], "view") else plain[i][1][:5], "batch", dist[i][:5])
