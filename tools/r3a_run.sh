#!/bin/bash
# from which corpus size the int8 filter scan pays (f32 / f16, D = 384), and the north-star shape 100M x 384 on ONE device: plain + filter
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
( for rows in 300000 600000 1200000 2500000 5000000; do
    VG_SCAN_FILTER_MIN_MB=0 python tools/tools_filter_selectivity.py --types f32,f16 --data gaussian --rows $rows --reps 20
  done ) 2>&1 | grep -v amdgpu.ids > $O/int8_filter_size_threshold.txt
timeout 900 python bench.py --rows 100000000 --steps 10 --warmup 2 --no-cpu-baseline --also filter 2>$O/bench100m.err | tail -1 > $O/c4_100Mx384_on_one_gpu_plain_and_filter.json
cat $O/int8_filter_size_threshold.txt; cut -c1-1500 $O/c4_100Mx384_on_one_gpu_plain_and_filter.json; tail -3 $O/bench100m.err
