cd /root/repo
mkdir -p gpurun_out/r4b
hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe 2>/dev/null && /tmp/valu_probe > gpurun_out/r4b/valu_probe.txt 2>&1
grep "waves/CU  4" gpurun_out/r4b/valu_probe.txt
tools/measure.sh r4b tests
python bench.py --workload stage > gpurun_out/r4b/bench_stage.json 2> gpurun_out/r4b/bench_stage.err; cut -c1-3000 gpurun_out/r4b/bench_stage.json; tail -2 gpurun_out/r4b/bench_stage.err
( echo "# f16 / bf16 plain kernels, default shapes"; python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 2,3 --filter 0
  echo "# f16 forced lpr16 u3"; VG_LPR_LOG2=4 VG_U=3 python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 2 --filter 0
  echo "# f16 forced lpr16 u4 (dim 512)"; python tools/tools_kernel_matrix.py --rows 8000000 --dims 512,768 --types 2,3 --filter 0 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4b/kernel_matrix_half.txt
cut -c1-260 gpurun_out/r4b/kernel_matrix_half.txt
python tools/shards_gather_bench.py > gpurun_out/r4b/shards_gather.json 2> gpurun_out/r4b/shards_gather.err; cat gpurun_out/r4b/shards_gather.json; tail -2 gpurun_out/r4b/shards_gather.err
