#!/bin/bash
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/probe_pmc_i8"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for grp in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum FETCH_SIZE"; do
  tag=$(echo $grp | tr " " "_" | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -f csv -d "$OUT/$tag" -o run -- python "$REPO/tools/tools_batch_bench.py" --type u8 --dim 768 --nq 1024 --metric 3 --reps 1 > "$OUT/$tag.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**/*counter_collection.csv"), recursive=True)):
    agg = {}
    for r in csv.DictReader(open(f)):
        k = (r.get("Kernel_Name", "")[:60], r.get("Counter_Name"))
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r.get("Counter_Value", 0))
    for (k, c), (n, v) in sorted(agg.items()):
        if "vg_batch_i8" in k:
            print("%-60s %-36s dispatches %3d  per dispatch %.6g" % (k, c, n, v / n))
PY
