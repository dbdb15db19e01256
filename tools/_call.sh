mkdir -p gpurun_out/r5m
timeout 900 python tools/shape_sweep.py --bytes 3.0e9 --rows 40000000 --cases 4:64,4:100,4:128,5:64 --repeat 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5m/int_short_rows_ab.txt
