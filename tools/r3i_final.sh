#!/bin/bash
# Round-2 closing measurement (gpurun): the GPU suite, the default bench line (+ rocprofv3 --kernel-trace --stats, + its own --pmc
# FETCH_SIZE run), the other workloads' lines, 100M x 384 on one device (plain + filter), the (type, metric) matrix.
tag=${1:-r3i}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1800 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1
tail -6 "$OUT/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cut -c1-300 "$OUT/bench_default.json"
for w in c1 c3b c5h; do timeout 600 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1; done > "$OUT/bench_lines_other_workloads.jsonl"
timeout 900 python bench.py --rows 100000000 --steps 10 --warmup 2 --no-cpu-baseline --also filter 2>/dev/null | tail -1 > "$OUT/c4_100Mx384_on_one_gpu_plain_and_filter.json"
( echo "# default path (int8 filter scans where served)"; python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3 --filter -1
  echo "# filter off (plain kernels)"; python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3,4,5 --filter 0 ) 2>&1 | grep -v amdgpu.ids > "$OUT/kernel_matrix.txt"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o run -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc" -o run -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_pmc.json" 2> "$OUT/rocprof_pmc.err"
cd "$REPO"
python tools/r2_summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
sed -n '/rocprofv3 --kernel-trace/,$p' "$OUT/summary.txt" | cut -c1-220 | head -40
find "$OUT" -name "*.csv" -size +8M -delete
