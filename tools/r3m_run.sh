cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3m
for div in 128 32 16 8; do
VG_SCAN_FILTER_PREPASS_DIV=$div timeout 600 python bench.py --also filter,c3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3m/bench_div$div.json
python - $div <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r3m/bench_div%s.json'%sys.argv[1])); f=d['also']['c3']['filter_scan']; g=d['filter_scan']
print("prepass 1/%s: c3 nibble: %.3f ms/query (kernel %.3f + prepass %.3f, %d evals) | c2 int8: %.3f ms/query (kernel %.3f + prepass %.3f, %d evals)"%(sys.argv[1], f['ms_per_step'],f['kernel_ms'],f['prepass_ms'],f['exact_evaluations_per_query'], g['ms_per_step'],g['kernel_ms'],g['prepass_ms'],g['exact_evaluations_per_query']))
PY
done | tee gpurun_out/r3m/prepass_div_sweep_c3_nibble.txt
