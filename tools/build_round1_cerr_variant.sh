#!/bin/bash
# Evidence build: libvectorgpu with round 1's UNSOUND filter constant (2^-8 (1 + 2^-8) instead of 2^-7 (1 + 2^-9)) in the
# single-query filter scan, to show that tests/test_gpu_filter_bound.py fails on it:
#   VG_LIB_PATH=$PWD/sqlite-vector_amd/libvectorgpu_round1cerr.so python -m pytest tests/test_gpu_filter_bound.py -m gpu -q
# Never used by the product (build.py does not know the macro).
set -e
cd "$(dirname "$0")/../sqlite-vector_amd"
python build.py >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -DVG_TEST_ROUND1_CERR -c csrc/vg_filter.hip -o build/vg_filter_round1cerr.o
objs=$(python -c "import build as b; print(' '.join('build/' + u[1] for u in b.HIP_UNITS if u[1] != 'vg_filter.hip.o'))")
hipcc --offload-arch=gfx950 -shared -fPIC -o libvectorgpu_round1cerr.so build/vg_filter_round1cerr.o $objs
echo built libvectorgpu_round1cerr.so
