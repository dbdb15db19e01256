cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3u
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/r3u/pytest_gpu_tail.txt; cat gpurun_out/r3u/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-also 2>/dev/null | tail -1 | cut -c1-420
