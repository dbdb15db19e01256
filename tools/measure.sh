#!/bin/bash
# One parameterised measurement script for a gpurun call (replaces the per-run r*_run.sh files of rounds 1-2).
#   usage: tools/measure.sh <tag> <step> [<step> ...]
#   steps: tests      pytest -m gpu (whole suite)                       -> pytest_gpu.txt
#          newtests   pytest -m gpu on the files named in $VG_TESTS     -> pytest_new.txt
#          bench      python bench.py (the driver's default command)    -> bench_default.json
#          stats      the same under rocprofv3 --kernel-trace --stats   -> stats/, summary.txt
#          pmc        the same under rocprofv3 --pmc FETCH_SIZE (own run, no other trace domain) -> pmc/, summary.txt,
#                     and profiles/pmc_traffic.json refreshed for THIS build's kernel sources (tools/summarize_profiles.py)
#          dist       the N-rank path on this 1-GPU box: 1 rank over RCCL (forced), 2 ranks sharing the device over gloo
#          others     the other workloads' lines (c1, c3b, c5h, stage)
#          matrix     tools_kernel_matrix.py: plain kernels + default path, all five types
#          stage      bench.py --workload stage (staging GB/s, minmax / quantize / int8-shadow kernels against the HBM peak)
#          shards     tools/shards_gather_bench.py: vg_shards' candidate gather, host vs RCCL, per query
#          probe      tools/valu_probe.hip: issue cost of the VALU instructions the f16 / bf16 kernels are made of
# Everything lands in gpurun_out/<tag>/; copy what is worth keeping into profiles/ (tools/summarize_profiles.py --keep does).
tag=${1:?tag}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
mkdir -p "$OUT"
cd "$REPO"
for step in "$@"; do
  echo "==== $step"
  case "$step" in
    tests)    timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -15 > "$OUT/pytest_gpu.txt"; cat "$OUT/pytest_gpu.txt" ;;
    newtests) timeout 1500 python -m pytest $VG_TESTS -m gpu -x -q 2>&1 | tail -40 > "$OUT/pytest_new.txt"; cat "$OUT/pytest_new.txt" ;;
    bench)    ( time timeout 900 python bench.py $VG_BENCH_ARGS ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -3 "$OUT/bench_default.err"; cut -c1-600 "$OUT/bench_default.json" ;;
    stats)    cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o run -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu-baseline $VG_BENCH_ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"; cd "$REPO"
              python tools/summarize_profiles.py "$OUT" > "$OUT/summary.txt" 2>&1; cut -c1-220 "$OUT/summary.txt" | head -60 ;;
    pmc)      cd /tmp; timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc" -o run -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --also filter,c3 $VG_BENCH_ARGS > "$OUT/bench_under_pmc.json" 2> "$OUT/rocprof_pmc.err"; cd "$REPO"
              python tools/summarize_profiles.py "$OUT" --update-traffic "$tag" > "$OUT/summary.txt" 2>&1; cut -c1-220 "$OUT/summary.txt" | head -80
              cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json" ;;
    dist)     VG_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline 2> "$OUT/dist_rccl_1rank.err" | tail -1 > "$OUT/dist_rccl_1rank.json"; cut -c1-400 "$OUT/dist_rccl_1rank.json"
              VG_BENCH_SHARE_DEVICES=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --cpu-sample-rows 200000 2> "$OUT/dist_selflaunch_2ranks_shared_device.err" | tail -1 > "$OUT/dist_selflaunch_2ranks_shared_device.json"; cut -c1-1200 "$OUT/dist_selflaunch_2ranks_shared_device.json"; tail -3 "$OUT/dist_selflaunch_2ranks_shared_device.err" ;;
    others)   for w in c1 c3b c5h stage; do timeout 600 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1; done > "$OUT/bench_lines_other_workloads.jsonl"; cut -c1-400 "$OUT/bench_lines_other_workloads.jsonl" ;;
    matrix)   ( echo "# default path"; python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3 --filter -1
                echo "# filter off (plain kernels)"; python tools/tools_kernel_matrix.py --rows 10000000 --dims 384 --types 1,2,3,4,5 --filter 0 ) 2>&1 | grep -v amdgpu.ids > "$OUT/kernel_matrix.txt"; cut -c1-300 "$OUT/kernel_matrix.txt" ;;
    stage)    timeout 600 python bench.py --workload stage 2> "$OUT/bench_stage.err" | tail -1 > "$OUT/bench_stage.json"; cut -c1-2500 "$OUT/bench_stage.json" ;;
    tlb)      # the plain f32 scan at 10M vs 100M rows: kernel time per batch order, then the translation / wait counters of the same dispatches
              rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
              for n in 10000000 100000000; do timeout 600 python tools/scan_size_probe.py --rows $n --orders 0,1,0,1 2>/dev/null | grep '^{'; done > "$OUT/scan_size_orders.jsonl"; cat "$OUT/scan_size_orders.jsonl"
              cd /tmp
              for n in 10000000 100000000; do for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" $VG_PMC_EXTRA_SETS; do
                  tagc=$(echo $set | cut -d' ' -f1)
                  timeout 600 rocprofv3 --pmc $set --kernel-trace -f csv -d "$OUT/tlb/n${n}_$tagc" -o run -- python "$REPO/tools/scan_size_probe.py" --rows $n --scans 4 --orders 0,1 > /dev/null 2> "$OUT/tlb_n${n}_$tagc.err" || echo "pmc set failed: $set"
              done; done
              cd "$REPO"; python tools/summarize_pmc_dir.py "$OUT/tlb" > "$OUT/tlb_summary.txt" 2>&1; cat "$OUT/tlb_summary.txt" ;;
    shards)   timeout 600 python tools/shards_gather_bench.py 2> "$OUT/shards_gather.err" | grep '^{' > "$OUT/shards_gather.json"; cat "$OUT/shards_gather.json" ;;
    probe)    hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe 2>/dev/null && /tmp/valu_probe > "$OUT/valu_probe.txt" 2>&1; grep "waves/CU  4" "$OUT/valu_probe.txt" ;;
    *)        echo "unknown step $step" ;;
  esac
done
find "$OUT" -name "*.csv" -size +8M -delete
