#!/bin/bash
# pipelined int8 batch kernel: parity first, then timings.  usage (through gpurun): tools/r2l_run.sh <tag> [types] [stages]
tag=${1:-r2l}
types=${2:-u8,u8s}
stages=${3:-200}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "quantized or staged_real or zero_queries or random_batches" > "$OUT/pytest_quantized.txt" 2>&1
tail -15 "$OUT/pytest_quantized.txt"
timeout 600 python tools/r2k_stage_sweep.py --types $types --stages $stages > "$OUT/sweep.jsonl" 2> "$OUT/sweep.err"
cat "$OUT/sweep.jsonl"; tail -3 "$OUT/sweep.err"
