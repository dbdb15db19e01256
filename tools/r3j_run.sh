cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_filter_bound.py -x -q -k "probes" 2>&1 | tail -12 > gpurun_out/r3j/pytest_probe.txt; cat gpurun_out/r3j/pytest_probe.txt
