cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_sql_extension.py -q -m gpu 2>&1 | grep -v amdgpu | tail -5 > gpurun_out/r3r/pytest_sql_ext.txt; cat gpurun_out/r3r/pytest_sql_ext.txt
