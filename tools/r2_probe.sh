#!/bin/bash
# Round-2 probe run (gpurun): suite, counter-evidence, half-type filter timings, int8 batch schedule variants
tag=${1:-r2b}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
if [ -f sqlite-vector_amd/libvectorgpu_round1cerr.so ]; then
  VG_LIB_PATH="$REPO/sqlite-vector_amd/libvectorgpu_round1cerr.so" timeout 600 python -m pytest tests/test_gpu_filter_bound.py -m gpu -q > "$OUT/filter_bound_on_round1_constant.txt" 2>&1
  tail -12 "$OUT/filter_bound_on_round1_constant.txt"
fi
( python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 2,3 --filter 0
  python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3 --filter 1
  python tools/tools_kernel_matrix.py --rows 5000000 --dims 768 --types 2,3 --filter 1
  python tools/tools_kernel_matrix.py --rows 20000000 --dims 128 --types 2,3 --filter 1 ) 2>&1 | grep -v amdgpu.ids > "$OUT/kernel_matrix_half_filter.txt"
cat "$OUT/kernel_matrix_half_filter.txt"
for v in "" i8ch2 i8ph2; do
  lib="$REPO/sqlite-vector_amd/libvectorgpu${v:+_$v}.so"
  [ -f "$lib" ] || continue
  for m in 3 4 1; do
    echo "== lib ${v:-default} metric $m"
    VG_LIB_PATH="$lib" timeout 300 python tools/tools_batch_bench.py --type u8 --dim 768 --nq 1024 --metric $m --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-420
  done
done > "$OUT/int8_batch_variants.txt" 2>&1
cat "$OUT/int8_batch_variants.txt"
