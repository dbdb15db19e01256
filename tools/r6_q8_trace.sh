#!/bin/bash
# Round 6: the dispatch timeline of the LAST batch of tools_q8_time.py (kernel name, duration, gap to the previous dispatch) per library build.
#   usage: tools/r6_q8_trace.sh <tag> <variant ...>
tag=${1:?tag}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/$tag"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD=${CMD:-"python $REPO/tools/tools_q8_time.py"}
for v in "$@"; do
  if [ "$v" = default ]; then lib="$REPO/sqlite-vector_amd/libvectorgpu.so"; else lib="$REPO/sqlite-vector_amd/libvectorgpu_q8_$v.so"; fi
  (cd /tmp; VG_LIB_PATH=$lib METRICS=${METRICS:-4} timeout 400 rocprofv3 --kernel-trace -f csv -d "$OUT/$v" -o run -- $CMD > "$OUT/$v.log" 2>&1 < /dev/null)
  grep "ms/batch" "$OUT/$v.log" | sed "s/^/$v: /"
  python - "$OUT/$v" "$v" <<'PY' | tee "$OUT/timeline_$v.txt"
import csv, glob, sys
d, v = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last batch = from the last vg_q8_query_prep_kernel pair on (two prep launches + rank per batch)
    idx = [i for i, r in enumerate(rows) if "vg_q8_query_prep" in r["Kernel_Name"]]
    if not idx:
        continue
    start = idx[-2]
    prev_end = None
    t0 = int(rows[start]["Start_Timestamp"])
    for r in rows[start:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("%-8s t=%8.1f us  dur %8.1f us  gap %6.1f  grid %6s wg %4s  %s" % (v, (s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r["Kernel_Name"][:60]))
        prev_end = e
PY
  rm -rf "$OUT/$v"
done
