#!/bin/bash
# pipelined int8 batch kernel v2: parity, timings, timing build
tag=${1:-r2n}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
L="$REPO/sqlite-vector_amd"
timeout 900 python -m pytest tests -m gpu -x -q -k "quantized or staged_real or zero_queries or random_batches" > "$OUT/pytest_quantized.txt" 2>&1
tail -15 "$OUT/pytest_quantized.txt"
timeout 600 python tools/r2k_stage_sweep.py --types u8,u8s --stages 200 > "$OUT/sweep.jsonl" 2> "$OUT/sweep.err"
cat "$OUT/sweep.jsonl"; tail -3 "$OUT/sweep.err"
if [ -f "$L/libvectorgpu_timing.so" ]; then
{
for spec in "768 4" "768 3" "128 3"; do
  set -- $spec
  VG_LIB_PATH="$L/libvectorgpu_timing.so" timeout 300 python tools/tools_i8_timing.py --dim $1 --metric $2 2>&1 | grep -v amdgpu.ids
done
} > "$OUT/timing.txt" 2>&1
cat "$OUT/timing.txt"
fi
