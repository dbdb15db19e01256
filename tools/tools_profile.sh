#!/bin/bash
# GPU-box profiling recipe (run through gpurun): rocprofv3 kernel-trace stats of bench.py, then a separate PMC pass
# for HBM traffic (FETCH_SIZE), per /opt/skills/guides/MI355X_MICROARCH.md.  Summaries land in gpurun_out/prof_*;
# the ones worth keeping are copied by hand into profiles/.
#   usage: ./tools_profile.sh <tag> [bench.py args...]
tag=${1:-r1}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/prof_$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
# pass 1: per-kernel time
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o run -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench_stats.log" 2>&1
# pass 2: HBM bytes actually fetched (own run: --pmc never together with the trace domains gpurun refuses)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc" -o run -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_pmc.log" 2>&1
cd "$REPO"
python - "$OUT" <<'EOF' > "$OUT/summary.txt" 2>&1
import csv, glob, os, sys, json
out = sys.argv[1]
def find(pat):
    r = glob.glob(os.path.join(out, pat), recursive=True)
    return r[0] if r else None
print("# bench line (under rocprofv3 --kernel-trace --stats)")
for l in open(os.path.join(out, "bench_stats.log")):
    if l.startswith("{"):
        print(l.strip())
st = find("stats/**/*kernel_stats.csv")
print("\n# rocprofv3 --kernel-trace --stats: per-kernel summary (%s)" % (os.path.relpath(st, out) if st else "MISSING"))
if st:
    rows = list(csv.DictReader(open(st)))
    for r in rows[:12]:
        print("%-72s calls %6s  total %12s ns  avg %12s ns  min %10s  max %10s  %6s%%" % (
            r.get("Name", "")[:72], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MinNs"),
            r.get("MaxNs"), r.get("Percentage")))
pm = find("pmc/**/*counter_collection.csv")
print("\n# rocprofv3 --pmc FETCH_SIZE (%s)" % (os.path.relpath(pm, out) if pm else "MISSING"))
if pm:
    agg = {}
    for r in csv.DictReader(open(pm)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        k = r.get("Kernel_Name", "")[:72]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r.get("Counter_Value", 0))
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        kb = v / n
        print("%-72s dispatches %4d  FETCH_SIZE/dispatch %.1f KB  -> x1024 = %.4g bytes ; x2 (gfx950 128B-request correction) = %.4g bytes" % (
            k, n, kb, kb * 1024, kb * 2048))
EOF
cat "$OUT/summary.txt"
