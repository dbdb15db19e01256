#!/bin/bash
# GPU-box experiment: A/B the scan kernel's launch shape on config C3 (10M x 768 u8 cosine).  Scratch tool.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/sweep_c3.log; : > $out
run() { echo "== $*" >> $out; env "$@" timeout 300 python bench.py --workload c3 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-28s scan %.3f ms  %.0f GB/s  frac %.3f  merge %.1f us  step %.3f ms p50 %.3f' % (r['kernel'], r['kernel_ms'], r['achieved'], r['frac'], r['merge_kernel_ms']*1e3, j['ms_per_step'], j['p50_query_latency_ms']))
" >> $out; }
run VG_X=0
run VG_LPR_LOG2=4 VG_U=3
run VG_LPR_LOG2=3 VG_U=6 VG_BLOCKS_PER_CU=2
run VG_LPR_LOG2=4 VG_U=3 VG_BLOCKS_PER_CU=2
run VG_LPR_LOG2=2 VG_U=12
run VG_LPR_LOG2=3 VG_U=8
run VG_LPR_LOG2=4 VG_U=4
run VG_LPR_LOG2=5 VG_U=2
run VG_NT=0
cat $out
