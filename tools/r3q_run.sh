cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3q
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -6 > gpurun_out/r3q/pytest_gpu_tail.txt; cat gpurun_out/r3q/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
