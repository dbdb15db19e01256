#!/bin/bash
# GPU-box experiment: A/B the scan kernel's launch shape on config C2 (10M x 384 f32 L2).  Scratch tool.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/sweep.log; : > $out
run() { echo "== $*" >> $out; env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-28s scan %.3f ms  %.0f GB/s  frac %.3f  merge %.1f us  step %.3f ms p50 %.3f' % (r['kernel'], r['kernel_ms'], r['achieved'], r['frac'], r['merge_kernel_ms']*1e3, j['ms_per_step'], j['p50_query_latency_ms']))
" >> $out; }
run VG_BLOCKS_PER_CU=4
run VG_BLOCKS_PER_CU=2
run VG_BLOCKS_PER_CU=3
run VG_BLOCKS_PER_CU=5
run VG_BLOCKS_PER_CU=8
run VG_BLOCKS_PER_CU=4 VG_LPR_LOG2=5 VG_U=3
run VG_BLOCKS_PER_CU=8 VG_LPR_LOG2=5 VG_U=3
run VG_BLOCKS_PER_CU=4 VG_LPR_LOG2=6 VG_U=2
run VG_BLOCKS_PER_CU=6 VG_LPR_LOG2=6 VG_U=2
run VG_BLOCKS_PER_CU=4 VG_LIB_PATH=$PWD/sqlite-vector_amd/libvectorgpu_nt.so
run VG_BLOCKS_PER_CU=8 VG_LIB_PATH=$PWD/sqlite-vector_amd/libvectorgpu_nt.so
run VG_BLOCKS_PER_CU=5 VG_LIB_PATH=$PWD/sqlite-vector_amd/libvectorgpu_nt.so
cat $out
