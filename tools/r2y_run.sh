#!/bin/bash
# int8 filter scan: pre-pass size and launch-shape sweep on N(0,1) 10M x 384, then the batch-policy test and the default bench line
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_scan.py -x -q -k "default_policy" 2>&1 | tail -15 > $O/pytest_policy.txt
( for div in 64 128 256 512; do echo "prepass 1/$div"; VG_SCAN_FILTER_PREPASS_DIV=$div python tools/tools_filter_selectivity.py --types f32 --data gaussian,clustered; done
  for shape in "3 3" "2 6" "4 2" "3 4" "2 4"; do set -- $shape; echo "lpr_log2 $1 U $2"; VG_FILTER_LPR_LOG2=$1 VG_FILTER_U=$2 python tools/tools_filter_selectivity.py --types f32 --data gaussian; done
  for bpc in 2 4 8; do echo "blocks per cu $bpc"; VG_BLOCKS_PER_CU=$bpc python tools/tools_filter_selectivity.py --types f32 --data gaussian; done
) 2>&1 | grep -v amdgpu.ids > $O/int8_filter_sweeps.txt
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_default_line.json
cat $O/pytest_policy.txt $O/int8_filter_sweeps.txt; cut -c1-300 $O/bench_default_line.json
