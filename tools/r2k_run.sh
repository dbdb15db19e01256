#!/bin/bash
# staged real passes: parity first, then the sweep.  usage (through gpurun): tools/r2k_run.sh <tag>
tag=${1:-r2k}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "staged_real_passes or two_pass" > "$OUT/pytest_staged.txt" 2>&1
tail -15 "$OUT/pytest_staged.txt"
timeout 600 python tools/r2k_stage_sweep.py --types u8,f16 --stages 0,400,200,150 > "$OUT/sweep.jsonl" 2> "$OUT/sweep.err"
cat "$OUT/sweep.jsonl"; tail -3 "$OUT/sweep.err"
timeout 900 python -m pytest tests -m gpu -x -q -k "batch" > "$OUT/pytest_batch.txt" 2>&1
tail -5 "$OUT/pytest_batch.txt"
