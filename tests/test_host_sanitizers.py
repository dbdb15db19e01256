"""CPU-only: the extension's HOST code under sanitizers (tools/asan_host_check.sh).  Without an engine: argument / option / JSON parsing
and every error path under ASan + UBSan.  Over a host-memory stub of the engine (tools/asan_stub_engine.c - test infrastructure, f32 / L2
only, never loaded by the product): the staging code - the single sqlite3_step loop, the parallel reader THREADS over key ranges and their
fallbacks (TEMP-table shadowing, a database held exclusively, a short BLOB inside a range), vector_quantize's staging in front of its
transaction - under ASan + UBSan and once more under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.sep in out and os.path.exists(out) else None


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_extension_host_code_is_clean_under_asan_ubsan_and_tsan():
    if not (_lib("libasan.so") and _lib("libubsan.so") and _lib("libtsan.so")):
        pytest.skip("sanitizer runtimes not installed")
    if not (os.path.exists("/usr/include/sqlite3ext.h") or os.path.exists("/root/reference/libs/sqlite3ext.h")):
        pytest.skip("no sqlite3ext.h to build the extension against")
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host_check.sh")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = p.stdout + p.stderr
    for bad in ("ERROR: AddressSanitizer", "runtime error", "WARNING: ThreadSanitizer", "Traceback"):
        assert bad not in out, out[-4000:]
    assert out.count("asan run done") == 4, out[-4000:]          # no engine | change tracking | staging under ASan | staging under TSan
