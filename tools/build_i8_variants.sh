#!/bin/bash
# Scratch: libvectorgpu variants that differ only in vg_batch_i8.hip compile-time switches, for A/B runs via VG_LIB_PATH.
#     tools/build_i8_variants.sh name -DVGI_TIMING=1 ...        (one translation unit for both of the product's: -DVGI_TU_ALL)
cd "$(dirname "$0")/../sqlite-vector_amd"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Icsrc -w -DVGI_TU_ALL "$@" -c csrc/vg_batch_i8.hip -o build/variant_i8_$name.o || exit 1
objs=""; for f in build/*.o; do case "$f" in build/vg_batch_i8*) ;; build/variant_*) ;; build/vg_batch_q8_*) ;; *) objs="$objs $f" ;; esac; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libvectorgpu_i8_$name.so $objs build/variant_i8_$name.o && echo built libvectorgpu_i8_$name.so
