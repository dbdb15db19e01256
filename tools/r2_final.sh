#!/bin/bash
# Round-2 full measurement (gpurun): suite, counter-evidence, default bench line + rocprofv3 passes, 100M x 384 on one GPU,
# every other workload's line.   usage: tools/r2_final.sh <tag>
tag=${1:-r2h}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1
tail -4 "$OUT/pytest_gpu.txt"
if [ -f sqlite-vector_amd/libvectorgpu_round1cerr.so ]; then
  VG_LIB_PATH="$REPO/sqlite-vector_amd/libvectorgpu_round1cerr.so" timeout 600 python -m pytest tests/test_gpu_filter_bound.py -m gpu -q -k "not guard" > "$OUT/filter_bound_on_round1_constant.txt" 2>&1
  tail -2 "$OUT/filter_bound_on_round1_constant.txt"
fi
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cut -c1-400 "$OUT/bench_default.json"
timeout 900 python bench.py --rows 100000000 --steps 10 --warmup 2 --no-also --no-cpu-baseline > "$OUT/c4_100Mx384_one_gpu_plain.json" 2>/dev/null
cut -c1-900 "$OUT/c4_100Mx384_one_gpu_plain.json"
for w in c1 c3 c5 c3b c5h c5f; do timeout 600 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1; done > "$OUT/bench_lines_other_workloads.jsonl"
cut -c1-300 "$OUT/bench_lines_other_workloads.jsonl"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o run -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc" -o run -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_pmc.json" 2> "$OUT/rocprof_pmc.err"
cd "$REPO"
python tools/r2_summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
sed -n '/rocprofv3 --kernel-trace/,$p' "$OUT/summary.txt" | cut -c1-260
find "$OUT" -name "*.csv" -size +8M -delete
