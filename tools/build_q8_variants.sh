#!/bin/bash
# Scratch: libvectorgpu variants that differ only in vg_batch_q8.hip's compile-time switches, for A/B runs via VG_LIB_PATH.
#     tools/build_q8_variants.sh name [-DVGQ_ABLATE=1 ...]      (ablation builds return wrong results: timing only)
cd "$(dirname "$0")/../sqlite-vector_amd"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Icsrc -w "$@" -c csrc/vg_batch_q8.hip -o build/vg_batch_q8_$name.o || exit 1
objs=""; for f in build/*.o; do case "$f" in build/vg_batch_q8*) ;; build/vg_batch_h_*variant*) ;; *) objs="$objs $f" ;; esac; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libvectorgpu_q8_$name.so $objs build/vg_batch_q8_$name.o && echo built libvectorgpu_q8_$name.so
