cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3t
timeout 1500 python -m pytest tests -q -m gpu -k "batch or default_policy or sql or golden" 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/r3t/pytest_batch_sql.txt; cat gpurun_out/r3t/pytest_batch_sql.txt
