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
rm -rf $out
