cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3p
timeout 900 python -m pytest tests/test_sql_extension.py -x -q -m gpu -k "through_the_filter" 2>&1 | grep -v amdgpu | tail -25 > gpurun_out/r3p/pytest_sql_filtered.txt; cat gpurun_out/r3p/pytest_sql_filtered.txt
