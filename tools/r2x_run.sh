#!/bin/bash
# int8 shadow copy for the f32 filter scans + default f32 batch filter policy: tests, selectivity, timing
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_filter_bound.py -x -q 2>&1 | tail -15 > $O/pytest_filter_bound.txt
timeout 600 python -m pytest tests/test_gpu_scan.py -x -q -k "filter" 2>&1 | tail -15 > $O/pytest_scan_filter.txt
for sh in int8 bf16; do
  VG_SCAN_FILTER_SHADOW=$sh timeout 600 python tools/tools_filter_selectivity.py --types f32 2>&1 | grep -v amdgpu.ids
done > $O/filter_selectivity_int8_vs_bf16.txt
cat $O/pytest_filter_bound.txt $O/pytest_scan_filter.txt $O/filter_selectivity_int8_vs_bf16.txt
