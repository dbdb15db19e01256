#!/bin/bash
# the nibble filter's probe; u8 selectivity in default mode (probe decides); whole suite
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter_bound.py -x -q -k "probes or nibble" 2>&1 | tail -15 > $O/pytest_probe.txt
( python tools/tools_filter_selectivity.py --types u8 --dim 768 --data gaussian,clustered ) 2>&1 | grep -v amdgpu.ids > $O/nibble_default_mode_probe.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu_tail.txt
cat $O/pytest_probe.txt $O/nibble_default_mode_probe.txt $O/pytest_gpu_tail.txt
