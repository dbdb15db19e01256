#!/bin/bash
tag=${1:-r2e}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
VG_LIB_PATH="$REPO/sqlite-vector_amd/libvectorgpu_round1cerr.so" timeout 600 python -m pytest tests/test_gpu_filter_bound.py -m gpu -q > "$OUT/filter_bound_on_round1_constant.txt" 2>&1
tail -3 "$OUT/filter_bound_on_round1_constant.txt"
( python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3 --filter 1
  python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 2,3 --filter 0
  python tools/tools_kernel_matrix.py --rows 5000000 --dims 768 --types 1,2,3 --filter 1
  python tools/tools_kernel_matrix.py --rows 20000000 --dims 128 --types 2,3 --filter 1
  python tools/tools_kernel_matrix.py --rows 20000000 --dims 128 --types 2,3 --filter 0 ) 2>&1 | grep -v amdgpu.ids > "$OUT/kernel_matrix_filter.txt"
cat "$OUT/kernel_matrix_filter.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cut -c1-600 "$OUT/bench_default.json"
