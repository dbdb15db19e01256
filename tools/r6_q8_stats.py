"""GPU-box tool (round 6): candidate statistics of vg_batch_q8_kernel per stage class (a -DVGQ_STATS=1 build, VG_LIB_PATH): per wave-tile, how many
tiles have a candidate lane, how many lanes are parked, how many registers pass the integer test, how many pairs leave."""
import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VG_BATCH_Q8"] = "1"
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, nq, k = int(os.environ.get("ROWS", "10000000")), int(os.environ.get("DIM", "384")), 1024, 20
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda")
for b in range(n // 500000):
    gen.manual_seed(42 * 100003 + b)
    t = torch.randn((500000, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), 500000, dim * 4); del t
qs = np.random.default_rng(44).standard_normal((nq, dim), dtype=np.float32)
lib = pkg.lib()
metric = int(os.environ.get("METRICS", "4"))
c.scan_topk_batch(metric, qs, k)
out = (ctypes.c_ulonglong * 32)()
lib.vg_batch_q8_stats(out, 1)
c.scan_topk_batch(metric, qs, k)
lib.vg_batch_q8_stats(out, 0)
print("metric", metric, "dim", dim, ": per stage class (tiles per partition) - wave-tiles | per wave-tile: tiles with a candidate lane, parked lanes, registers past the integer test, pairs")
for b, name in enumerate(("< 16", "16-63", "64-255", "256-447", "448-999", ">= 1000")):
    wt, slow, lanes, cand, pairs = [int(out[5 * b + i]) for i in range(5)]
    if wt:
        print("  %-8s wave-tiles %9d | slow %.4f  lanes %.4f  registers %.4f  pairs %.4f   (pairs per lane %.2f)" % (name, wt, slow / wt, lanes / wt, cand / wt, pairs / wt, pairs / max(1, lanes)))
if out[31]:
    print("  last stage: a query's own threshold term lies %.4f (relative) below its set's loosest, mean over %d queries" % (out[30] / 1e6 / out[31], out[31]))
