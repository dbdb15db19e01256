#!/bin/bash
# Scratch: libvectorgpu variants that differ only in vg_batch_hl.hip (compile-time switches), one row length / metric only, for A/B runs
# via VG_LIB_PATH.      tools/build_long_variants.sh name [-DVGHL_ABLATE=1 ...]        (needs an up-to-date build/ of everything else)
cd "$(dirname "$0")/../sqlite-vector_amd"
name=$1; shift
pids=""
for tu in 0 1 2 3 4 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Icsrc -Wno-unused-value -Wno-unused-result -DVGHL_TU=$tu -DVGHL_ONLY_NTBP=${NTBP:-24} -DVGHL_ONLY_DOT "$@" -c csrc/vg_batch_hl.hip -o build/vg_batch_hl_${tu}_$name.o &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
objs=""; for f in build/*.o; do case $f in build/vg_batch_hl_*) ;; *) objs="$objs $f";; esac; done
for tu in 0 1 2 3 4 5; do objs="$objs build/vg_batch_hl_${tu}_$name.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libvectorgpu_$name.so $objs && echo built libvectorgpu_$name.so
rm -f build/vg_batch_hl_*_$name.o
