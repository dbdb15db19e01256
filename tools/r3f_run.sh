#!/bin/bash
# grouped exact evaluation (up to 64 / xlpr candidate rows per wavefront at once): parity + what unselective data costs now
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_filter_bound.py tests/test_gpu_scan.py -x -q -k "filter or nibble or int8 or patch" 2>&1 | tail -6 > $O/pytest_filters.txt
( VG_SCAN_FILTER_N4=1 VG_SCAN_FILTER_NO_GUARD=1 python tools/tools_filter_selectivity.py --types u8,i8 --dim 768 --data gaussian,clustered
  VG_SCAN_FILTER_N4=1 VG_SCAN_FILTER_NO_GUARD=1 python tools/tools_filter_selectivity.py --types u8 --dim 384 --data gaussian
  python tools/tools_filter_selectivity.py --types f32,f16 --data gaussian,clustered,near ) 2>&1 | grep -v amdgpu.ids > $O/filter_selectivity_grouped_exact.txt
cat $O/pytest_filters.txt $O/filter_selectivity_grouped_exact.txt
