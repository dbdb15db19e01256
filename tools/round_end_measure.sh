cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/pytest_gpu.txt
for w in c1 c2 c3 c5 c3b c5h; do python bench.py --workload $w 2>/dev/null | tail -1; done > gpurun_out/final/bench_lines.jsonl
python bench.py --workload c2 --rows 100000000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/c4_one_gpu.json
VG_LIB_PATH=/root/repo/sqlite-vector_amd/libvectorgpu_timing.so python tools/tools_half_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/half_timing.txt
VG_LIB_PATH=/root/repo/sqlite-vector_amd/libvectorgpu_timing.so python tools/tools_half_timing.py --nq 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/final/half_timing.txt
( for m in 4 3 1; do python tools/tools_batch_bench.py --type f16 --dim 384 --nq 1024 --metric $m; done; python tools/tools_batch_bench.py --type bf16 --dim 384 --nq 1024,256,64 --metric 3; python tools/tools_batch_bench.py --type f16 --dim 128 --nq 1024 --metric 1; python tools/tools_batch_bench.py --type bf16 --dim 512 --rows 5000000 --nq 1024 --metric 4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/final/half_batch.jsonl
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/final/prof_c5h -- python bench.py --workload c5h --steps 5 --warmup 1 > /dev/null 2>&1
cat gpurun_out/final/pytest_gpu.txt; cut -c1-250 gpurun_out/final/bench_lines.jsonl; cat gpurun_out/final/half_timing.txt
