cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_sql_extension.py -x -q -s -m gpu -k "fuzz_random_statement" 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/r3o/pytest_track_fuzz.txt; cat gpurun_out/r3o/pytest_track_fuzz.txt
