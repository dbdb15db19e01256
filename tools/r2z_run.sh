#!/bin/bash
# int8 shadow copy for f32 AND f16 / bf16 corpora: parity tests over both shadow settings, selectivity + timing
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2z; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_filter_bound.py -x -q 2>&1 | tail -15 > $O/pytest_filter_bound.txt
timeout 1500 python -m pytest tests/test_gpu_scan.py -x -q -k "filter" 2>&1 | tail -15 > $O/pytest_scan_filter.txt
for sh in int8 rows; do
  VG_SCAN_FILTER_SHADOW=$sh timeout 900 python tools/tools_filter_selectivity.py --types f16,bf16 --data gaussian,clustered 2>&1 | grep -v amdgpu.ids
done > $O/filter_selectivity_half_int8_vs_rows.txt
cat $O/pytest_filter_bound.txt $O/pytest_scan_filter.txt $O/filter_selectivity_half_int8_vs_rows.txt
