#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
( for rows in 150000 300000 600000 1000000 1200000; do
    VG_SCAN_FILTER_MIN_MB=0 python tools/tools_filter_selectivity.py --types f32,f16 --data gaussian --rows $rows --reps 20
  done ) 2>&1 | grep -v amdgpu.ids > $O/int8_filter_size_threshold_grouped.txt
cat $O/int8_filter_size_threshold_grouped.txt
