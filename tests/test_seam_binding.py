"""INTEGRATION.md section B, executed (VERDICT r4 missing #3): the REFERENCE's own sqlite-vector.c - argument checks, JSON parsing, slot
sort, vector_quantize, everything - with include/vectorgpu.h bound at its run-callback seam (sqlite-vector.c:183-184, 2116, 2179, 1397),
built out of tree by oracle/seam/make_seam.py into oracle/_ref/seam/vector.so, and the golden SQL cases of the reference run through THAT
library.  The drop-in claim of the boundary is then a test, not a paragraph."""
import os
import sqlite3

import numpy as np
import pytest

import datagen as dg
from test_sql_extension import _check_vs_golden, load_table, mg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SEAM = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "seam", "vector")


@pytest.fixture(scope="module")
def seam_path(orc):
    if not os.path.exists(SEAM + ".so"):
        pytest.skip("oracle/_ref/seam/vector.so not built (needs /root/reference at build time; it travels with the snapshot)")
    return SEAM


def connect(path):
    db = sqlite3.connect(":memory:", isolation_level=None)
    db.enable_load_extension(True)
    db.load_extension(path)
    return db


@pytest.mark.parametrize("case", mg.SQL_SCAN_CASES, ids=[c[0] for c in mg.SQL_SCAN_CASES])
def test_reference_host_code_with_the_gpu_bound_at_its_seam_full_scan(seam_path, case):
    name, vt, metric, n, dim, k, seed, low = case
    sql = np.load(os.path.join(HERE, "golden", "sql.npz"))
    rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
    q = dg.query(vt, dim, seed + 1, low_entropy=low)
    db = connect(seam_path)
    assert "gfx950" in db.execute("SELECT vector_backend()").fetchone()[0]        # the patched vector_backend (:2549)
    load_table(db, rows, vt, metric)
    got = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    _check_vs_golden(got, sql["avx2/%s/rowids" % name], sql["avx2/%s/dist" % name], exact=vt in (dg.U8, dg.I8))
    if vt == dg.F32:                                                             # the reference's own JSON path in front of the seam
        js = "[" + ",".join(repr(float(x)) for x in q) + "]"
        assert db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (js, k)).fetchall() == got
    best = got[0][0]                                                             # freshness at the seam: stamps moved -> re-staged
    db.execute("DELETE FROM t WHERE id=?", (best,))
    got2 = db.execute("SELECT rowid FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    assert best not in [g[0] for g in got2] and len(got2) == k


@pytest.mark.parametrize("case", mg.SQL_QUANT_CASES, ids=[c[0] for c in mg.SQL_QUANT_CASES])
def test_reference_host_code_with_the_gpu_bound_at_its_seam_quantized_scan(seam_path, case):
    name, vt, qopt, n, dim, k, seed, nonneg = case
    sql = np.load(os.path.join(HERE, "golden", "sql.npz"))
    rows = dg.corpus(vt, n, dim, seed)
    if nonneg:
        rows = np.abs(rows)
    q = dg.query(vt, dim, seed + 1)
    db = connect(seam_path)
    load_table(db, rows, vt, dg.COSINE)
    db.execute("SELECT vector_quantize('t','v',?)", ("qtype=%s" % qopt,)) if qopt else db.execute("SELECT vector_quantize('t','v')")   # the reference's own CPU quantizer
    want_ids, want_bits = sql["avx2/%s/rowids" % name], sql["avx2/%s/dist" % name]
    got_cpu = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()     # not preloaded: the reference's disk loop
    _check_vs_golden(got_cpu, want_ids, want_bits, exact=True)
    db.execute("SELECT vector_quantize_preload('t','v')")                         # the patched preload: records -> vg_corpus_append_records
    got = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()         # vQuantRunGPU
    _check_vs_golden(got, want_ids, want_bits, exact=True)
    assert got == got_cpu
