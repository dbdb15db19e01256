export VG_TESTS="tests/test_gpu_filter_bound.py tests/test_gpu_scan.py tests/test_tie_order.py tests/test_gpu_fuzz.py"
bash tools/measure.sh r5j newtests stage pmc bench
