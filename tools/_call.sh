mkdir -p gpurun_out/r5e
( timeout 900 python -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "minmax or quantize or half_types or f32_all or int8_bit" 2>&1 | tail -6 ) > gpurun_out/r5e/pytest.txt; cat gpurun_out/r5e/pytest.txt
( timeout 600 python -m pytest tests/test_sql_extension.py -m gpu -x -q -k "quantize" 2>&1 | tail -4 ) > gpurun_out/r5e/pytest_sql.txt; cat gpurun_out/r5e/pytest_sql.txt
bash tools/measure.sh r5e stage 2>&1 | cut -c1-1800
timeout 600 python tools/shape_sweep.py --cases 3:384,2:384 --repeat 1 2>&1 | grep -v amdgpu.ids | grep "default" | tee gpurun_out/r5e/half_default.txt
