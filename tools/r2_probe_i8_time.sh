#!/bin/bash
# timings only: int8 batch kernel variants next to the default build.   usage: tools/r2_probe_i8_time.sh tag variant...
tag=$1; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
mkdir -p "$OUT"
cd "$REPO"
for l in "" "$@"; do
  lp="$REPO/sqlite-vector_amd/libvectorgpu${l:+_$l}.so"
  [ -f "$lp" ] || continue
  for spec in "768 3 1024" "768 4 1024" "128 3 1024" "768 3 256"; do
    set -- $spec
    echo "== lib ${l:-default} dim $1 metric $2 nq $3"
    VG_LIB_PATH="$lp" timeout 300 python tools/tools_batch_bench.py --type u8 --dim $1 --nq $3 --metric $2 --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done
done > "$OUT/timings.txt" 2>&1
cat "$OUT/timings.txt"
