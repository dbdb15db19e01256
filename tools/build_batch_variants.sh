#!/bin/bash
# Scratch: build libvectorgpu variants that differ only in vg_batch.hip compile-time switches, for A/B runs via VG_LIB_PATH.
#   tools/build_batch_variants.sh name "-DVGB_ABLATE=7" ...
cd "$(dirname "$0")/../sqlite-vector_amd"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result "$@" -c csrc/vg_batch.hip -o build/vg_batch_$name.o || exit 1
objs=""; for f in vg_api vg_select vg_quant vg_shards; do objs="$objs build/$f.hip.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libvectorgpu_$name.so $objs build/vg_batch_$name.o && echo built libvectorgpu_$name.so
