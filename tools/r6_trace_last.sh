#!/bin/bash
# the dispatch timeline of the LAST batch of a command (kernels after the last occurrence of $MARK in kernel names) under rocprofv3 --kernel-trace
# usage: MARK=<substring of the batch's first kernel> tools/r6_trace_last.sh <tag> <command ...>
tag=${1:?tag}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$REPO/gpurun_out/$tag"; mkdir -p "$OUT"; export TMPDIR=/tmp
(cd /tmp; timeout 600 rocprofv3 --kernel-trace -f csv -d "$OUT/tr" -o run -- "$@" > "$OUT/cmd.log" 2>&1 < /dev/null)
grep "ms/batch\|ms_per" "$OUT/cmd.log" | head -5
python - "$OUT/tr" "${MARK:-prep}" <<'PY' | tee "$OUT/timeline.txt"
import csv, glob, sys
d, mark = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
    if not idx: print("mark not found"); continue
    # the last batch: from the first marked kernel of the last run of marks
    start = idx[-1]
    while start - 1 in idx: start -= 1
    import os
    if os.environ.get("LASTN"): start = max(0, len(rows) - int(os.environ["LASTN"]))
    t0, prev = int(rows[start]["Start_Timestamp"]), None
    for r in rows[start:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("t=%8.1f us dur %8.1f gap %6.1f grid %8s wg %4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), r["Kernel_Name"][:70]))
        prev = e
PY
rm -rf "$OUT/tr"
