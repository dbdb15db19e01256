#!/bin/bash
# Build the extension host (plain C) with AddressSanitizer + UBSan and drive its argument / option / JSON parsing, the
# host quantization passes and every error path without the GPU engine (VECTORGPU_LIB points nowhere).  CPU only.
#   tools/asan_host_check.sh        -> prints "asan run done" and no sanitizer report when clean
set -e
cd "$(dirname "$0")/.."
inc=/usr/include; [ -f $inc/sqlite3ext.h ] || inc=/root/reference/libs
out=$(mktemp -d)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -I$inc -Iinclude -Isqlite-vector_amd/ext -o $out/vector.so sqlite-vector_amd/ext/vector_ext.c -ldl -lm -lpthread 2>/dev/null
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 VECTORGPU_LIB=/nonexistent \
    python tools/asan_host_drive.py $out/vector 2>&1 | grep -E "ERROR: AddressSanitizer|runtime error|asan run done" || true
# ... and the staging code - single loop, parallel reader THREADS and their fallbacks, vector_quantize's staging in front of its transaction -
# over a host-memory stub of the engine (tools/asan_stub_engine.c: test infrastructure, f32 / L2 only), under ASan + UBSan and under TSan
gcc -O1 -g -fPIC -shared -o $out/stub.so tools/asan_stub_engine.c -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 VECTORGPU_LIB=$out/stub.so \
    python tools/asan_staging_drive.py $out/vector 2>&1 | grep -E "ERROR: AddressSanitizer|runtime error|asan run done|Traceback|Error" || true
gcc -O1 -g -fsanitize=thread -fno-omit-frame-pointer -fPIC -shared -I$inc -Iinclude -Isqlite-vector_amd/ext -o $out/vector_tsan.so sqlite-vector_amd/ext/vector_ext.c -ldl -lm -lpthread 2>/dev/null
mv $out/vector_tsan.so $out/tsan_dir_vector.so 2>/dev/null; mkdir -p $out/tsan; mv $out/tsan_dir_vector.so $out/tsan/vector.so
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="report_bugs=1 exitcode=0" VECTORGPU_LIB=$out/stub.so \
    python tools/asan_staging_drive.py $out/tsan/vector 2>&1 | grep -E "WARNING: ThreadSanitizer|vext_|stage_|asan run done|Traceback" | head -40 || true
rm -rf $out
