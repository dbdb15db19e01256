#!/bin/bash
# GPU-box: bench C5 with each libvectorgpu_<variant>.so built by tools/build_batch_variants.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; out=gpurun_out/c5_variants.log; : > $out
for v in "" "$@"; do
  lib=sqlite-vector_amd/libvectorgpu${v:+_$v}.so
  echo "== ${v:-default}" >> $out
  VG_LIB_PATH=$PWD/$lib timeout 300 python bench.py --workload c5 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%s  kernel %.2f ms  %.1f TF  frac %.3f' % (r['kernel'], r['kernel_ms'], r['achieved'], r['frac']))
" >> $out
done
cat $out
