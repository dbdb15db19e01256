#!/bin/bash
# Round-2 third-session measurement (gpurun): default bench line (+ rocprofv3 --kernel-trace --stats, + its own --pmc FETCH_SIZE run),
# the other workloads' lines, the (type, metric) matrix at 10M x 384 through the default (int8 filter) path and with the filter off,
# other dims through the int8 filter.     usage: tools/r3c_final.sh <tag>
tag=${1:-r3c}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cut -c1-400 "$OUT/bench_default.json"
for w in c1 c3b c5h; do timeout 600 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1; done > "$OUT/bench_lines_other_workloads.jsonl"
cut -c1-300 "$OUT/bench_lines_other_workloads.jsonl"
( echo "# default path (int8 filter scans where served)"; python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3 --filter -1
  echo "# filter off (plain kernels)"; python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3,4,5 --filter 0
  echo "# other dims, default path, 8 GB of rows each"; python tools/tools_kernel_matrix.py --bytes 8e9 --dims 64,128,768,1024,1536 --types 1,2 --filter -1 ) 2>&1 | grep -v amdgpu.ids > "$OUT/kernel_matrix.txt"
cat "$OUT/kernel_matrix.txt" | cut -c1-330
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o run -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc" -o run -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_pmc.json" 2> "$OUT/rocprof_pmc.err"
cd "$REPO"
python tools/r2_summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
sed -n '/rocprofv3 --kernel-trace/,$p' "$OUT/summary.txt" | cut -c1-260
find "$OUT" -name "*.csv" -size +8M -delete
