cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3k
python tools/tools_row_maintenance.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3k/row_maintenance_10Mx384.txt; cat gpurun_out/r3k/row_maintenance_10Mx384.txt
timeout 600 python bench.py --also filter,c3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3k/bench_c2_c3.json
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3k/bench_c2_c3.json')); print(json.dumps(d['also']['c3'].get('nibble_filter_probe')))
PY
