#!/bin/bash
# half-precision batch kernel on the tile-major copy: parity, then timings with and without it
tag=${1:-r2u}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "half or f32_long or f32_768 or bf16_filter or staged_real or random_batches or batch" > "$OUT/pytest_half.txt" 2>&1
tail -8 "$OUT/pytest_half.txt"
for tm in 1 0; do
  echo "== VG_BATCH_TILE_MAJOR=$tm"
  VG_BATCH_TILE_MAJOR=$tm timeout 600 python tools/r2k_stage_sweep.py --types f16,bf16 --stages 400 2>&1 | grep -v amdgpu.ids
done > "$OUT/sweep.txt" 2>&1
cat "$OUT/sweep.txt"
