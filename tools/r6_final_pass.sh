#!/bin/bash
# GPU-box: the round's closing measurement pass on ONE box - whole GPU suite, default bench, the same under rocprofv3 (stats, then FETCH_SIZE),
# the N-rank path, the other workloads' lines, and the three batched PMC passes whose traffic entries bench.py reads.
#     tools/r6_final_pass.sh <tag>      -> gpurun_out/<tag>/
tag=${1:?tag}
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
python - <<'PY'
import time, sys
sys.path.insert(0, ".")
t0 = time.time()
import __graft_entry__ as g
pkg = g.load_package(); pkg.lib()
t1 = time.time()
import numpy as np
c = pkg.Corpus(pkg.F32, 384); c.append(np.zeros((64, 384), np.float32)); c.scan_topk(pkg.L2, np.zeros(384, np.float32), 5)
print("library load %.3f s, first corpus + first scan (code objects inflated and loaded) %.3f s" % (t1 - t0, time.time() - t1))
PY
tools/measure.sh $tag tests bench stats pmc dist others stage
OUT="$REPO/gpurun_out/$tag"
PMC_KEEP=${tag}_c5_default_path_int8_filter_pmc.txt PMC_ENTRY=batch_path7_f32_dot_1024q_384@10000000 tools/pmc_batch.sh "$OUT/pmc_c5" > "$OUT/c5_default_path_int8_filter_pmc.txt" 2>&1
PMC_KEEP=${tag}_long_rows_int8_filter_pmc_10m_1536.txt PMC_ENTRY=batch_path7_f32_dot_1024q_1536@10000000 tools/pmc_batch.sh "$OUT/pmc_long" --dim 1536 > "$OUT/long_rows_int8_filter_pmc_10m_1536.txt" 2>&1
PMC_KEEP=${tag}_c3b_uint8_batch_pmc.txt PMC_ENTRY=batch_path2_u8_cosine_1024q_768@10000000 tools/pmc_batch.sh "$OUT/pmc_c3b" --dim 768 --type u8 --metric 3 > "$OUT/c3b_uint8_batch_pmc.txt" 2>&1
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
rm -rf "$OUT"/pmc_c5/p* "$OUT"/pmc_long/p* "$OUT"/pmc_c3b/p*
tail -12 "$OUT/c5_default_path_int8_filter_pmc.txt"
# the bench once more, now that the traffic entries carry this build's hash
python bench.py > "$OUT/bench_default_with_traffic.json" 2> /dev/null; cut -c1-900 "$OUT/bench_default_with_traffic.json"
