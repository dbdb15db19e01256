cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3w2
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r3w2/bench_default.json; cut -c1-200 gpurun_out/r3w2/bench_default.json
