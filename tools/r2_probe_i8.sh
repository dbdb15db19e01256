#!/bin/bash
# probe: an int8 batch kernel variant (VG_LIB_PATH) - bit-exactness tests, then timings next to the default build
v=${1:-i8async}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/probe_$v"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
lib="$REPO/sqlite-vector_amd/libvectorgpu_$v.so"
VG_LIB_PATH="$lib" timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q -x -k "quantized or batch" > "$OUT/pytest.txt" 2>&1
tail -4 "$OUT/pytest.txt"
for l in "" "$v"; do
  lp="$REPO/sqlite-vector_amd/libvectorgpu${l:+_$l}.so"
  for spec in "768 3 1024" "768 4 1024" "768 1 1024" "128 3 1024" "768 3 256" "1024 3 1024"; do
    set -- $spec
    echo "== lib ${l:-default} dim $1 metric $2 nq $3"
    VG_LIB_PATH="$lp" timeout 300 python tools/tools_batch_bench.py --type u8 --dim $1 --nq $3 --metric $2 --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-330
  done
done > "$OUT/timings.txt" 2>&1
cat "$OUT/timings.txt"
