#!/bin/bash
# int8 batch kernel: variant libraries (libvectorgpu_<name>.so) next to the default build, then the timing build
tag=${1:-r2o}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
L="$REPO/sqlite-vector_amd"
{
for v in "" "$@"; do
  lp="$L/libvectorgpu${v:+_$v}.so"
  [ -f "$lp" ] || continue
  echo "== lib ${v:-default}"
  VG_LIB_PATH="$lp" timeout 300 python tools/r2k_stage_sweep.py --types u8,u8s --stages 200 --nq 1024 2>&1 | grep -v amdgpu.ids
done
for tl in timing locktiming g0timing; do
if [ -f "$L/libvectorgpu_$tl.so" ]; then
echo "== $tl"
for spec in "768 4" "768 3"; do
  set -- $spec
  VG_LIB_PATH="$L/libvectorgpu_$tl.so" timeout 300 python tools/tools_i8_timing.py --dim $1 --metric $2 2>&1 | grep -v amdgpu.ids
done
fi
done
} > "$OUT/variants.txt" 2>&1
cat "$OUT/variants.txt"
