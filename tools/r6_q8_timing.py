"""GPU-box tool (round 6): shader-clock ticks per wavefront class of vg_batch_q8_kernel (a -DVGQ_TIMING=1 build, VG_LIB_PATH): where a wave-tile's cycles go."""
import ctypes, os, sys, time, numpy as np
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
for i in range(2): c.scan_topk_batch(metric, qs, k)
out = (ctypes.c_ulonglong * 16)()
lib.vg_batch_q8_timing(out, 1)
t0 = time.perf_counter()
c.scan_topk_batch(metric, qs, k)
ms = (time.perf_counter() - t0) * 1e3
lib.vg_batch_q8_timing(out, 0)
name = os.environ.get("VG_LIB_PATH", "default").split("/")[-1]
print(name, "ms/batch %.3f" % ms)
for o, cls in ((0, "waves 0-3"), (8, "waves 4-7")):
    T = max(1, out[o + 4])
    print("  %s: wave-tiles %d | per wave-tile: loop %.0f  k-loop %.0f  boundary %.0f  wait+barrier %.0f (of it the DMA wait %.0f) ticks" % (cls, out[o + 4], out[o] / T, out[o + 1] / T, out[o + 2] / T, out[o + 3] / T, out[o + 5] / T))
