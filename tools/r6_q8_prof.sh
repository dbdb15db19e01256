#!/bin/bash
# Round 6: per-kernel time of one workload under rocprofv3 --kernel-trace --stats, for several library builds, in one gpurun call.
#   usage: tools/r6_q8_prof.sh <tag> <variant ...>     ("default" = libvectorgpu.so; others libvectorgpu_q8_<name>.so);  CMD overrides the workload
tag=${1:?tag}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/$tag"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD=${CMD:-"python $REPO/tools/tools_q8_time.py"}
for v in "$@"; do
  if [ "$v" = default ]; then lib="$REPO/sqlite-vector_amd/libvectorgpu.so"; else lib="$REPO/sqlite-vector_amd/libvectorgpu_q8_$v.so"; fi
  (cd /tmp; VG_LIB_PATH=$lib METRICS=${METRICS:-4} timeout 400 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/$v" -o run -- $CMD > "$OUT/$v.log" 2>&1 < /dev/null)
  grep "ms/batch" "$OUT/$v.log" | sed "s/^/$v: /"
  python - "$OUT/$v" "$v" <<'PY' | tee -a "$OUT/summary.txt"
import csv, glob, sys
d, v = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:int(8)]:
        print("%-10s %-78s calls %6s total_ms %9.3f avg_us %9.2f" % (v, r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  rm -rf "$OUT/$v"
done
