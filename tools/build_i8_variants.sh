#!/bin/bash
# Scratch: libvectorgpu variants that differ only in vg_batch_i8.hip compile-time switches, for A/B runs via VG_LIB_PATH.
#     tools/build_i8_variants.sh name -DVGI_PRIO=1 ...        (VGI_SRC=other_file.hip: another source in csrc/)
cd "$(dirname "$0")/../sqlite-vector_amd"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Icsrc -Wno-unused-value -Wno-unused-result -DVGI_TU_ALL "$@" -c csrc/${VGI_SRC:-vg_batch_i8.hip} -o build/vg_batch_i8_$name.o || exit 1
objs="build/vg_batch_h_bf16.o build/vg_batch_h_bound.o build/vg_batch_h_f32.o"; for f in vg_api vg_corpus vg_batch_api vg_select vg_batch vg_quant vg_shards vg_batch_h vg_multi vg_reforder vg_filter; do objs="$objs build/$f.hip.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libvectorgpu_$name.so $objs build/vg_batch_i8_$name.o && echo built libvectorgpu_$name.so
