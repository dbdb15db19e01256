"""GPU-box tool (round 6): per-tile timestamps of the eight wavefronts of ONE workgroup of vg_batch_q8_kernel's last stage (a -DVGQ_TRACE=1
build): when each wavefront's k loop begins / ends, when its boundary ends, when it is past the group's wait + barrier"""
import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VG_BATCH_Q8"] = "1"
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, dim, nq, k = 10_000_000, 384, 1024, 20
c = pkg.Corpus(pkg.F32, dim, capacity=n)
gen = torch.Generator(device="cuda")
for b in range(n // 500000):
    gen.manual_seed(42 * 100003 + b)
    t = torch.randn((500000, dim), generator=gen, device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    c.append_device(t.data_ptr(), 500000, dim * 4); del t
qs = np.random.default_rng(44).standard_normal((nq, dim), dtype=np.float32)
lib = pkg.lib()
for i in range(2): c.scan_topk_batch(4, qs, k)
NT = 96
out = (ctypes.c_ulonglong * (8 * NT * 4))()
lib.vg_batch_q8_trace(out, 1)
c.scan_topk_batch(4, qs, k)
lib.vg_batch_q8_trace(out, 0)
a = np.array(out[:], dtype=np.int64).reshape(8, NT, 4)
t0 = a[a > 0].min()
a = np.where(a > 0, a - t0, -1)
print("# tile | per wave: k-begin k-end b-end past-barrier (cycles from the first record)")
for ti in range(0, 24):
    print("tile %2d" % ti)
    for w in range(8):
        print("   wave %d: %6d %6d %6d %6d   k %5d  b %5d  wait %5d" % (w, a[w, ti, 0], a[w, ti, 1], a[w, ti, 2], a[w, ti, 3], a[w, ti, 1] - a[w, ti, 0], a[w, ti, 2] - a[w, ti, 1], a[w, ti, 3] - a[w, ti, 2]))
per = (a[:, NT - 1, 3] - a[:, 0, 0]) / NT
print("cycles per tile over %d tiles, by wave:" % NT, per.round(0).tolist())
