#!/bin/bash
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/r2i"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_tie_order.py tests/test_sql_extension.py -m gpu -x -q 2>&1 | tail -2
VG_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_forced_dist_1rank.json" 2> "$OUT/bench_forced_dist.err"
tail -c 1500 "$OUT/bench_forced_dist_1rank.json"; tail -3 "$OUT/bench_forced_dist.err"
VG_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --rows 12500000 --steps 5 --warmup 2 --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-700
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['also']['c3']['tie_order'])"
