"""summarise the rocprofv3 passes of tools/r2_measure.sh: per-kernel average durations (--kernel-trace --stats) and
FETCH_SIZE per dispatch (--pmc FETCH_SIZE, its own run; x1024 bytes/KB, x2 gfx950 128-B request correction)"""
import csv
import glob
import os
import sys

out = sys.argv[1]


def find(pat):
    r = glob.glob(os.path.join(out, pat), recursive=True)
    return r[0] if r else None


for name in ("bench_default.json", "bench_under_rocprof.json"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        print("# %s" % name)
        for l in open(p):
            if l.startswith("{"):
                print(l.strip())
st = find("stats/**/*kernel_stats.csv")
print("\n# rocprofv3 --kernel-trace --stats: per-kernel summary (%s)" % (os.path.relpath(st, out) if st else "MISSING"))
if st:
    for r in list(csv.DictReader(open(st)))[:24]:
        print("%-90s calls %6s  total %12s ns  avg %12s ns  min %10s  max %10s  %6s%%" % (
            r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MinNs"),
            r.get("MaxNs"), r.get("Percentage")))
pm = find("pmc/**/*counter_collection.csv")
print("\n# rocprofv3 --pmc FETCH_SIZE (%s)" % (os.path.relpath(pm, out) if pm else "MISSING"))
if pm:
    agg = {}
    for r in csv.DictReader(open(pm)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        k = r.get("Kernel_Name", "")[:90]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r.get("Counter_Value", 0))
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        kb = v / n
        print("%-90s dispatches %4d  FETCH_SIZE/dispatch %.1f KB  -> x1024 x2 (gfx950 128B-request correction) = %.5g bytes" % (k, n, kb, kb * 2048))
