#!/bin/bash
# GPU-box tool: PMC passes over one batch workload (tools_batch_bench.py: 1024 queries x 10M rows, warm-up + 1 timed batch) - per kernel,
# summed over its dispatches: MFMA instructions, VALU instructions, cycles the MFMA pipe was busy, active cycles, FETCH_SIZE.
#     tools/pmc_batch.sh <out-dir> [--dim 384 --type f32 ...]     (every pass is a rocprofv3 --pmc run of its own, --kernel-trace only)
export TMPDIR=/tmp
OUT=$1; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"; export VG_REPO="$REPO"
mkdir -p "$OUT"
OUT="$(cd "$OUT" && pwd)"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  i=$((i+1))
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc $set -f csv -d "$OUT/p$i" -o run -- python "$REPO/tools/tools_batch_bench.py" --rows 10000000 --nq 1024 --reps 1 "$@" > "$OUT/p$i.log" 2>&1 < /dev/null)
done
python - "$OUT" "$*" <<'PY'
import csv, glob, sys, collections
out, what = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not n.startswith("void vg_") and not n.startswith("vg_"):
            continue
        a = agg.setdefault(n.split("(")[0][:70], {"disp": set(), "c": collections.OrderedDict(), "ns": {}})
        a["disp"].add((f, r["Dispatch_Id"]))
        a["ns"][(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a["c"][r["Counter_Name"]] = a["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("# tools/pmc_batch.sh %s: tools_batch_bench.py --rows 10000000 --nq 1024 --reps 1, TWO batches (warm-up + 1 rep) per pass; counters summed over both" % what)
for n, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]["ns"].values())):
    c = a["c"]
    if sum(a["ns"].values()) < 50_000:
        continue
    print("%s   (dispatch rows: %d, %.3f ms in all passes)" % (n, len(a["disp"]), sum(a["ns"].values()) / 1e6))
    for k, v in c.items():
        print("   %-28s %.6g" % (k, v))
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("   -> MFMA pipe busy %.1f %% of (1024 SIMDs x active cycles);  VALU per MFMA instruction %.2f" % (
            100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] * 8 / (1024.0 * c["GRBM_GUI_ACTIVE"]), c.get("SQ_INSTS_VALU", 0.0) / max(c.get("SQ_INSTS_MFMA", 1.0), 1.0)))
    if c.get("FETCH_SIZE"):
        print("   -> HBM bytes (FETCH_SIZE KB x 1024 x 2) per batch: %.4g GB" % (c["FETCH_SIZE"] * 1024 * 2 / 2 / 1e9))
# HBM bytes of ONE batch = every launch of one vg_scan_topk_batch call (the filter, exact-evaluation, merge, query-image kernels; NOT the
# one-off passes that build the shadow copies in front of the first batch), both batches of the pass / 2.  PMC_ENTRY=<key> writes it into
# profiles/pmc_traffic.json (bench.py: batch_traffic_entry) stamped with the hash of the batch kernels' sources.
import json, os
tot_kb = sum(a["c"].get("FETCH_SIZE", 0.0) for n, a in agg.items() if ("vg_batch_" in n or "vg_q8_" in n))
per_batch = tot_kb * 1024 * 2 / 2
print("# all batch kernels: HBM bytes per batch %.4g GB" % (per_batch / 1e9))
entry = os.environ.get("PMC_ENTRY")
if entry and tot_kb > 0:
    root = os.path.dirname(os.path.dirname(os.path.abspath(out))) if False else os.environ.get("VG_REPO", ".")
    sys.path.insert(0, root)
    import bench
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    tab = json.load(open(path)) if os.path.exists(path) else {}
    tab[entry] = {"bytes_per_batch": int(round(per_batch)), "kernel_source_hash": bench.batch_kernel_source_hash(),
                  "source": "profiles/%s (tools/pmc_batch.sh %s: rocprofv3 --pmc FETCH_SIZE over two batches, its own pass; KB x 1024 x 2 = gfx950 128-B request correction; every launch of one batch, the one-off shadow-copy passes excluded)" % (os.environ.get("PMC_KEEP", "?"), what)}
    json.dump(tab, open(path, "w"), indent=1)
    print("# profiles/pmc_traffic.json: %s = %d bytes per batch" % (entry, int(round(per_batch))))
PY
