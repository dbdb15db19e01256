import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch; torch.cuda.init()
import __graft_entry__ as g
import datagen as dg
pkg = g.load_package()
os.environ['VG_SCAN_FILTER_MIN_MB'] = '0'
bad = 0
for dim in (384, 100, 768, 33, 1024):
    n = 200_003
    rows = dg.corpus(dg.F32, n, dim, 11 + dim)
    _, edge = dg.edge_rows(dg.F32, dim, 12 + dim)
    rows[500:500 + len(edge)] = edge
    rows[70000] = rows[17]
    c = pkg.Corpus(pkg.F32, dim); c.append(rows)
    for metric in (dg.L2, dg.SQUARED_L2, dg.DOT):
        for qi in range(6):
            q = dg.query(dg.F32, dim, 100 + qi) if qi else rows[17].copy()
            for k in (1, 20, 64):
                os.environ["VG_SCAN_FILTER"] = "1"; i1, d1 = c.scan_topk(metric, q, k)
                os.environ["VG_SCAN_FILTER"] = "0"; i0, d0 = c.scan_topk(metric, q, k)
                ok = len(i1) == len(i0) and len(set(i1.tolist()) ^ set(i0.tolist())) <= 2
                if ok and metric != dg.DOT: ok = np.allclose(d1, d0, rtol=1e-5, atol=1e-6)
                if not ok:
                    bad += 1; print("MISMATCH", dim, metric, qi, k, i1[:6], i0[:6], d1[:4], d0[:4])
    c.close()
print("mismatches:", bad)
