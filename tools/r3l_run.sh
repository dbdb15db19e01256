cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3l
timeout 600 python bench.py --also filter,c3 --no-cpu-baseline 2>gpurun_out/r3l/err.txt | tail -1 > gpurun_out/r3l/bench_c2_c3.json
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3l/bench_c2_c3.json')); c3=d['also']['c3']; print(c3['ms_per_step'], c3['roofline']['frac']); print(json.dumps(c3.get('nibble_filter_probe'))); print(json.dumps(c3.get('filter_scan'))[:1200])
PY
tail -2 gpurun_out/r3l/err.txt
