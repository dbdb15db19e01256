#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu_tail.txt
timeout 600 python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_default.json
cat $O/pytest_gpu_tail.txt; cut -c1-300 $O/bench_default.json; tail -2 $O/bench.err
