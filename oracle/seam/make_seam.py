#!/usr/bin/env python3
"""INTEGRATION.md section B, executed: the REFERENCE's own sqlite-vector.c with the C-ABI bound at its seam.

    python oracle/seam/make_seam.py            ->  oracle/_ref/seam/vector.so   (+ the spliced source next to it, for inspection)

Reads /root/reference/src/sqlite-vector.c where it lies, applies the anchored edits below (each anchor is a few tokens of the reference's
text - a function name, a struct's closing line - never a copy of its code), writes the result OUT OF TREE (oracle/_ref/ is git-ignored,
travels to the GPU box) and compiles it with the reference's other sources against include/vectorgpu.h, linked to
sqlite-vector_amd/libvectorgpu.so.  Test infrastructure: tests/test_seam_binding.py loads the result; the product never does."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REF", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref", "seam")


def once(text, pattern, repl, what, flags=0):
    new, n = re.subn(pattern, repl, text, count=1, flags=flags)
    if n != 1:
        raise SystemExit("make_seam: anchor not found - %s" % what)
    return new


def main():
    src_path = os.path.join(REF, "src", "sqlite-vector.c")
    if not os.path.exists(src_path):
        print("make_seam: no reference tree at %s - nothing built" % REF)
        return 0
    s = open(src_path).read()
    # 1. table_context (sqlite-vector.c:125-137) gets the staged corpora + the stamps they were staged at
    s = once(s, r"(\n\s*int\s+precounter;\s*\n)(\}\s*table_context;)",
             r"\1    void *gpu_full, *gpu_quant; sqlite3_int64 gpu_dv, gpu_ch;      /* seam: include/vectorgpu.h */\n\2\n"
             r"static void seam_drop_quant (table_context *t);\nstatic void seam_preload_to_gpu (table_context *t, const void *records, int counter);", "table_context's last field")
    # 2. the seam's functions, in front of vFullScanSortSlots (:2051): behind every typedef they use
    s = once(s, r"(\nstatic int vFullScanSortSlots\s*\()", r'\n#include "vectorgpu_seam.inc"\n\1', "vFullScanSortSlots")
    # 3. the run callback of vector_full_scan (:2116)
    s = once(s, r'("vector_full_scan",\s*)vFullScanRun(\s*,)', r"\1vFullScanRunGPU\2", "vector_full_scan's run callback")
    # 4. vQuantRun (:2179): the preloaded branch goes to the device when the preload reached it
    s = once(s, r"(\n\s*)(if \(c->table->preloaded\) \{\s*\n\s*int rc = vQuantRunMemory)",
             r"\1if (c->table->gpu_quant) { int rcg = vQuantRunGPU(c, v, qtype, dimension); if (v) sqlite3_free(v); return rcg; }\1\2", "vQuantRun's preloaded branch")
    # 5. vector_quantize_preload (:1397-1398): the assembled records also go to HBM; dropping the host copy drops the device copy (:1352, :1513)
    s = once(s, r"(t_ctx->precounter = counter;\s*\n)", r"\1    seam_preload_to_gpu(t_ctx, buffer, counter);\n", "vector_quantize_preload's hand-over")
    s, n = re.subn(r"(if \(t_ctx->preloaded\) \{\s*\n)", r"\1        seam_drop_quant(t_ctx);\n", s)
    if n < 2:
        raise SystemExit("make_seam: anchor not found - the two places that free `preloaded`")
    # 6. vector_backend (:2549)
    s = once(s, r"(static void vector_backend\s*\([^)]*\)\s*\{\s*\n\s*sqlite3_result_text\(context,\s*)distance_backend_name", r"\1vg_backend_name()", "vector_backend")
    os.makedirs(OUT, exist_ok=True)
    spliced = os.path.join(OUT, "sqlite-vector-seam.c")
    open(spliced, "w").write(s)
    lib_dir = os.path.join(ROOT, "sqlite-vector_amd")
    srcs = [spliced] + [os.path.join(REF, "src", f) for f in ("distance-cpu.c", "distance-avx2.c", "distance-sse2.c", "distance-neon.c")]
    cmd = ["gcc", "-O3", "-fPIC", "-w", "-mavx2", "-shared", "-I" + os.path.join(REF, "src"), "-I" + os.path.join(REF, "libs"), "-I" + os.path.join(ROOT, "include"),
           "-I" + HERE, "-o", os.path.join(OUT, "vector.so")] + srcs + ["-L" + lib_dir, "-lvectorgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/root/repo/sqlite-vector_amd", "-lm"]
    if not os.path.exists(os.path.join(lib_dir, "libvectorgpu.so")):
        print("make_seam: sqlite-vector_amd/libvectorgpu.so is not built yet - nothing built")
        return 0
    subprocess.run(cmd, check=True)
    print("make_seam: built", os.path.join(OUT, "vector.so"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
