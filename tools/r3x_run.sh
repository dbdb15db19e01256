cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3x
VG_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-also 2>gpurun_out/r3x/err.txt | tail -1 > gpurun_out/r3x/bench_forced_dist_1rank.json
cut -c1-700 gpurun_out/r3x/bench_forced_dist_1rank.json; tail -3 gpurun_out/r3x/err.txt | cut -c1-200
