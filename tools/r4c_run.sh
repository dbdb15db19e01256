cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_filter_bound.py tests/test_sql_extension.py tests/test_tie_order.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_subset.txt; cat $O/pytest_subset.txt
timeout 600 python bench.py --no-cpu-baseline --also filter,matrix > $O/bench_matrix.json 2> $O/bench_matrix.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4c/bench_matrix.json') if l.startswith('{')][0])
print('headline', d['roofline']['frac'], d['ms_per_step'], 'filter', d['filter_scan']['ms_per_step'], d['filter_scan']['kernel_ms'], d['filter_scan']['prepass_ms'])
for r in d['also']['kernel_matrix']['rows']: print(r.get('dtype'), r.get('metric'), r.get('kernel'), r.get('kernel_ms'), r.get('frac'))
PY
for div in 256 512; do VG_SCAN_FILTER_PREPASS_DIV=$div timeout 300 python bench.py --no-cpu-baseline --also filter --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); f=d['filter_scan']
print('prepass div $div', f['ms_per_step'], f['kernel_ms'], f['prepass_ms'], f['exact_evaluations_per_query'])"; done | tee $O/prepass_div.txt
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/$O/stage_stats -o run -- python /root/repo/bench.py --workload stage > /root/repo/$O/bench_stage_rocprof.json 2> /root/repo/$O/stage_rocprof.err; cd /root/repo
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4c/stage_stats/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]: print(r['Name'][:80], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
cut -c1-200 $O/bench_stage_rocprof.json | head -3
VG_Q8_TWO_READS=1 python bench.py --workload stage 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('two reads q8', d['quantize']['q8_shadow'])"
find $O -name "*.csv" -size +4M -delete
