#!/bin/bash
# the nibble filter for uint8 / int8 corpora: parity tests, selectivity + timing at 10M x 768 (config C3's shape) and 10M x 384
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_filter_bound.py -x -q -k "nibble" 2>&1 | tail -25 > $O/pytest_nibble.txt
( python tools/tools_filter_selectivity.py --types u8,i8 --dim 768 --data gaussian,clustered
  python tools/tools_filter_selectivity.py --types u8 --dim 384 --data gaussian ) 2>&1 | grep -v amdgpu.ids > $O/nibble_filter_selectivity.txt
timeout 600 python bench.py --also filter,c3 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_c2_c3_with_filters.json
cat $O/pytest_nibble.txt $O/nibble_filter_selectivity.txt; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r3d/bench_c2_c3_with_filters.json'))
c3=d['also']['c3']; print('c3 plain', c3['ms_per_step'], c3['roofline']['frac'], c3['roofline']['kernel']); print(json.dumps(c3.get('filter_scan'))[:900])
PY
