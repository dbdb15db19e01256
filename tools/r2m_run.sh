#!/bin/bash
# pipelined int8 batch kernel: where the time goes (timing build) and what the DMA issue / the tests cost (ablation builds)
tag=${1:-r2m}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
L="$REPO/sqlite-vector_amd"
{
for spec in "768 4" "768 3" "128 3"; do
  set -- $spec
  VG_LIB_PATH="$L/libvectorgpu_timing.so" timeout 300 python tools/tools_i8_timing.py --dim $1 --metric $2 2>&1 | grep -v amdgpu.ids
done
for v in nodma nojudge; do
  echo "== $v"
  VG_LIB_PATH="$L/libvectorgpu_$v.so" timeout 300 python tools/r2k_stage_sweep.py --types u8,u8s --stages 200 --nq 1024 2>&1 | grep -v amdgpu.ids
done
} > "$OUT/timing.txt" 2>&1
cat "$OUT/timing.txt"
