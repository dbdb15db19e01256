#!/bin/bash
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/probe_tiled"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_sql_extension.py -m gpu -q -x -k "quantized or batch or quant" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
for l in "" "$@"; do
  lp="$REPO/sqlite-vector_amd/libvectorgpu${l:+_$l}.so"
  [ -f "$lp" ] || continue
  for spec in "768 3 1024" "768 4 1024" "768 1 1024" "128 3 1024" "384 3 1024" "768 3 256" "1024 3 1024" "1536 3 1024"; do
    set -- $spec
    echo "== lib ${l:-default} dim $1 metric $2 nq $3"
    VG_LIB_PATH="$lp" timeout 300 python tools/tools_batch_bench.py --type u8 --dim $1 --nq $3 --metric $2 --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done
done > "$OUT/timings.txt" 2>&1
grep -o '== lib.*\|"kernel_ms": [0-9.]*' "$OUT/timings.txt" | paste - -
if [ -f "$REPO/sqlite-vector_amd/libvectorgpu_timing.so" ]; then
  export VG_LIB_PATH=$REPO/sqlite-vector_amd/libvectorgpu_timing.so
  for m in 3 4; do python tools/tools_i8_timing.py --dim 768 --metric $m 2>&1 | grep -v amdgpu; done
fi
