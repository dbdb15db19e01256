#!/bin/bash
# GPU-box profiling of the batched MFMA kernel (config C5): kernel-trace stats, then PMC passes for MFMA busy cycles.
tag=${1:-r1_c5}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/prof_$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o run -- python "$REPO/tools/tools_batch_bench.py" --nq 1024 --reps 2 "$@" > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$OUT/pmc1" -o run -- python "$REPO/tools/tools_batch_bench.py" --nq 1024 --reps 1 "$@" > "$OUT/bench_pmc1.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -f csv -d "$OUT/pmc2" -o run -- python "$REPO/tools/tools_batch_bench.py" --nq 1024 --reps 1 "$@" > "$OUT/bench_pmc2.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import csv, glob, os, sys
out = sys.argv[1]
def find(pat):
    r = glob.glob(os.path.join(out, pat), recursive=True); return r[0] if r else None
print("# bench lines (under rocprofv3 --kernel-trace --stats)")
for l in open(os.path.join(out, "bench_stats.log")):
    if l.startswith("{"): print(l.strip())
st = find("stats/**/*kernel_stats.csv")
print("\n# per-kernel summary")
if st:
    for r in list(csv.DictReader(open(st)))[:8]:
        print("%-60s calls %5s total %12s ns avg %12s ns  %6s%%" % (r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
for p in ("pmc1", "pmc2"):
    pm = find(p + "/**/*counter_collection.csv")
    print("\n# PMC pass %s (%s)" % (p, "ok" if pm else "MISSING - see bench_%s.log" % p))
    if not pm: continue
    agg = {}
    for r in csv.DictReader(open(pm)):
        if "vg_batch_kernel" not in r["Kernel_Name"]: continue
        a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        agg.setdefault("_dur_ns", [0, 0.0]); 
    for k, (n, v) in sorted(agg.items()):
        if n: print("  %-32s dispatches %3d  mean %.6g" % (k, n, v / n))
    kt = find(p + "/**/*kernel_trace.csv")
    if kt:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "vg_batch_kernel" in r["Kernel_Name"]]
        if d: print("  kernel duration (ns) per dispatch: %s" % d)
PY
cat "$OUT/summary.txt"
