#!/bin/bash
# Scratch: libvectorgpu variants that differ only in vg_batch_h.hip (compile-time switches or another source file), for
# A/B runs via VG_LIB_PATH.      tools/build_half_variants.sh name [source.hip] [-DVGH_ABLATE=1 ...]
cd "$(dirname "$0")/../sqlite-vector_amd"
name=$1; shift
src=csrc/vg_batch_h.hip
if [ -n "$1" ] && [ "${1#-}" = "$1" ]; then src=$1; shift; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Icsrc -Wno-unused-value -Wno-unused-result -DVGH_TU_ALL "$@" -c $src -o build/vg_batch_h_$name.o || exit 1
objs="build/vg_batch_i8_pre.o"; for f in vg_api vg_corpus vg_batch_api vg_select vg_batch vg_quant vg_shards vg_batch_i8 vg_multi vg_reforder vg_scan_ex vg_filter; do objs="$objs build/$f.hip.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libvectorgpu_$name.so $objs build/vg_batch_h_$name.o && echo built libvectorgpu_$name.so
