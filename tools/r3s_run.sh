cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3s
timeout 1200 python -m pytest tests -q -m gpu -k "batch" 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/r3s/pytest_batch.txt; cat gpurun_out/r3s/pytest_batch.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
