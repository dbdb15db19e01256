#!/bin/bash
# row maintenance (patch / delete / find), tracked changes through the update hook, then the whole GPU suite
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scan.py -x -q -k "patch_and_delete" 2>&1 | tail -25 > $O/pytest_patch_delete.txt
timeout 900 python -m pytest tests/test_sql_extension.py -x -q -m gpu -k "tracked or incremental" 2>&1 | tail -25 > $O/pytest_tracked.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu_tail.txt
cat $O/pytest_patch_delete.txt $O/pytest_tracked.txt $O/pytest_gpu_tail.txt
