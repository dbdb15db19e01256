#!/bin/bash
# Round-2 GPU-box recipe (run through gpurun): parity suite, the round-1-constant counter-evidence, the bench line, and
# the rocprofv3 passes the bench line's numbers are checked against.  usage: tools/r2_measure.sh <tag> [quick]
tag=${1:-r2a}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
if [ -f sqlite-vector_amd/libvectorgpu_round1cerr.so ]; then
  VG_LIB_PATH="$REPO/sqlite-vector_amd/libvectorgpu_round1cerr.so" timeout 600 python -m pytest tests/test_gpu_filter_bound.py -m gpu -q > "$OUT/filter_bound_on_round1_constant.txt" 2>&1
  tail -3 "$OUT/filter_bound_on_round1_constant.txt"
fi
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cut -c1-1500 "$OUT/bench_default.json"
[ "$2" = "quick" ] && exit 0
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o run -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc" -o run -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_pmc.json" 2> "$OUT/rocprof_pmc.err"
cd "$REPO"
python tools/r2_summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the merge-back small: the raw traces can be large
find "$OUT" -name "*.csv" -size +8M -delete
