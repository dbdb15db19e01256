#!/bin/bash
# Round 6: A/B of vg_batch_q8 builds in one gpurun call.  usage: tools/r6_q8_ab.sh <tag> <variant names ...>  ("default" = libvectorgpu.so)
tag=${1:?tag}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
OUT="$REPO/gpurun_out/$tag"; mkdir -p "$OUT"
libof() { if [ "$1" = default ]; then echo "$REPO/sqlite-vector_amd/libvectorgpu.so"; else echo "$REPO/sqlite-vector_amd/libvectorgpu_q8_$1.so"; fi; }
if [ -n "$VG_TESTS" ]; then timeout 900 python -m pytest $VG_TESTS -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/pytest.txt"; fi
for round in 1 2; do
  for v in "$@"; do
    VG_LIB_PATH=$(libof $v) METRICS=${METRICS:-4,1,3} timeout 300 python tools/tools_q8_time.py 2>&1 | grep "ms/batch" | sed "s/^/round $round $v: /" | tee -a "$OUT/q8_short.txt"
  done
done
if [ -n "$LONG" ]; then
  for v in "$@"; do
    VG_LIB_PATH=$(libof $v) timeout 400 python bench.py --workload c5l --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'c5l ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])" | tee -a "$OUT/q8_long.txt"
  done
fi
