mkdir -p gpurun_out/r5c
timeout 600 python tools/shape_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c/shape_sweep.txt
