cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests -q -m gpu -k "patch_and_delete or tracked or incremental" 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/r3v/pytest_maintenance.txt; cat gpurun_out/r3v/pytest_maintenance.txt
python tools/tools_row_maintenance.py --rows 4000000 2>&1 | grep -v amdgpu | grep "patch" > gpurun_out/r3v/patch_timings_4M.txt; cat gpurun_out/r3v/patch_timings_4M.txt
