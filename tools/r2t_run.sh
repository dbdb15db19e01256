#!/bin/bash
# round-2 int8 batch kernel, final schedule: whole GPU suite, batch timings of several shapes, PMC passes
tag=${1:-r2t}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1
tail -6 "$OUT/pytest_gpu.txt"
timeout 900 python tools/r2k_stage_sweep.py --types u8,i8,u8s,f16,bf16 --stages 200,0 > "$OUT/sweep.jsonl" 2> "$OUT/sweep.err"
cat "$OUT/sweep.jsonl"
{
for spec in "u8 384 3" "u8 1024 3" "u8 1536 3" "i8 768 1"; do
  set -- $spec
  echo "== $1 dim $2 metric $3"
  timeout 300 python tools/tools_batch_bench.py --type $1 --dim $2 --nq 1024 --metric $3 --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-400
done
} > "$OUT/shapes.txt" 2>&1
cat "$OUT/shapes.txt"
timeout 600 tools/tools_profile_batch_pmc.sh u8 768 4 "$OUT/pmc_u8_dot" > "$OUT/pmc_u8_dot.txt" 2>&1
timeout 600 tools/tools_profile_batch_pmc.sh u8 768 3 "$OUT/pmc_u8_cos" > "$OUT/pmc_u8_cos.txt" 2>&1
grep -A30 "false, true>" "$OUT/pmc_u8_dot.txt" | grep -- "->\|vg_batch" ; grep -- "->\|vg_batch" "$OUT/pmc_u8_cos.txt"
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
