"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE (oracle/_ref, built from /root/reference by
oracle/Makefile).  Run in the build container:  python tests/golden/make_golden.py

Inputs are never stored: they are regenerated from seeds by tests/datagen.py.  Outputs stored:
  kernels.npz   float32 bit patterns of dispatch_distance_table[metric][type](query,row,dim) for both reference
                builds ("CPU" = distance-cpu.c, "AVX2" = distance-avx2.c), random and edge-case rows.
  sql.npz       results of the reference extension driven through SQL: vector_full_scan / vector_quantize /
                vector_quantize_scan (rowids, distance bits, quantization parameters, quantized bytes).
"""
import os
import sqlite3
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import datagen as dg  # noqa: E402
from oracle import orc  # noqa: E402

KERNEL_CASES = [(35, 48, 101), (384, 32, 102), (768, 16, 103)]      # (dim, n_rows, seed)
EDGE_DIMS = (13, 384)

TYPE_OPT = {dg.F32: "FLOAT32", dg.F16: "FLOAT16", dg.BF16: "FLOATB16", dg.U8: "UINT8", dg.I8: "INT8"}
DIST_OPT = {dg.L2: "L2", dg.SQUARED_L2: "SQUARED_L2", dg.COSINE: "COSINE", dg.DOT: "DOT", dg.L1: "L1"}

# (name, type, metric, n, dim, k, seed, low_entropy)
SQL_SCAN_CASES = [
    ("c1_f32_l2", dg.F32, dg.L2, 10000, 384, 20, 42, False),       # BASELINE config #1
    ("f32_cos", dg.F32, dg.COSINE, 1500, 64, 20, 43, False),
    ("f32_dot", dg.F32, dg.DOT, 1500, 64, 20, 44, False),
    ("f32_l1", dg.F32, dg.L1, 1500, 100, 7, 45, False),
    ("f16_l2", dg.F16, dg.L2, 1200, 96, 20, 46, False),
    ("bf16_cos", dg.BF16, dg.COSINE, 1200, 96, 20, 47, False),
    ("u8_l2_ties", dg.U8, dg.L2, 2000, 32, 20, 48, True),
    ("i8_dot_ties", dg.I8, dg.DOT, 2000, 32, 20, 49, True),
    ("u8_cos", dg.U8, dg.COSINE, 2000, 768, 20, 50, False),
]
# (name, source type, qtype option, n, dim, k, seed, nonneg)
SQL_QUANT_CASES = [
    ("q_f32_auto_u8", dg.F32, None, 3000, 768, 20, 60, True),      # BASELINE config #3 scaled down
    ("q_f32_auto_s8", dg.F32, None, 2000, 128, 20, 61, False),
    ("q_f16_u8", dg.F16, "UINT8", 1000, 64, 10, 62, False),
    ("q_bf16_s8", dg.BF16, "INT8", 1000, 64, 10, 63, False),
]


def gen_kernels():
    out = {}
    for which in ("cpu", "avx2"):
        ref = orc.RefKernels(which)
        for vt in dg.ALL_TYPES:
            for (dim, n, seed) in KERNEL_CASES:
                rows = dg.corpus(vt, n, dim, seed)
                q = dg.query(vt, dim, seed + 7)
                for m in dg.ALL_METRICS:
                    out["%s/rand/%s/%s/%d" % (which, dg.TYPE_NAMES[vt], dg.METRIC_NAMES[m], dim)] = \
                        ref.scan(m, vt, q, rows).view(np.uint32)
            for dim in EDGE_DIMS:
                _, rows = dg.edge_rows(vt, dim, 1000 + dim)
                for qi, q in enumerate(dg.edge_queries(vt, dim, 2000 + dim)):
                    for m in dg.ALL_METRICS:
                        out["%s/edge/%s/%s/%d/%d" % (which, dg.TYPE_NAMES[vt], dg.METRIC_NAMES[m], dim, qi)] = \
                            ref.scan(m, vt, q, rows).view(np.uint32)
    return out


def connect(which):
    db = sqlite3.connect(":memory:", isolation_level=None)
    db.enable_load_extension(True)
    db.load_extension(orc.ref_extension_path(which))
    return db


def load(db, rows, vt, metric):
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(rows.shape[0])])
    db.execute("SELECT vector_init('t', 'v', ?)",
               ("type=%s,dimension=%d,distance=%s" % (TYPE_OPT[vt], rows.shape[1], DIST_OPT[metric]),))


def gen_sql():
    out = {}
    for which in ("cpu", "avx2"):
        for (name, vt, metric, n, dim, k, seed, low) in SQL_SCAN_CASES:
            db = connect(which)
            rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
            q = dg.query(vt, dim, seed + 1, low_entropy=low)
            load(db, rows, vt, metric)
            got = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
            out["%s/%s/rowids" % (which, name)] = np.array([g[0] for g in got], dtype=np.int64)
            out["%s/%s/dist" % (which, name)] = np.array([g[1] for g in got], dtype=np.float32).view(np.uint32)
            db.close()
        for (name, vt, qopt, n, dim, k, seed, nonneg) in SQL_QUANT_CASES:
            db = connect(which)
            rows = dg.corpus(vt, n, dim, seed)
            if nonneg:
                rows = np.abs(rows)
            q = dg.query(vt, dim, seed + 1)
            load(db, rows, vt, dg.COSINE)
            if qopt:
                db.execute("SELECT vector_quantize('t','v',?)", ("qtype=%s" % qopt,))
            else:
                db.execute("SELECT vector_quantize('t','v')")
            meta = dict(db.execute("SELECT key, value FROM _sqliteai_vector WHERE tblname='t'").fetchall())
            out["%s/%s/qparams" % (which, name)] = np.array([meta["qtype"], meta["qscale"], meta["qoffset"]],
                                                            dtype=np.float64)
            blob = b"".join(r[0] for r in db.execute("SELECT data FROM vector0_t_v ORDER BY rowid1").fetchall())
            rec = np.frombuffer(blob, dtype=np.uint8).reshape(n, 8 + dim)
            # keep the fixture small: first 64 quantized rows verbatim + a byte-sum per row for the rest
            out["%s/%s/qhead" % (which, name)] = rec[:64, 8:].copy()
            out["%s/%s/qrowsum" % (which, name)] = rec[:, 8:].astype(np.uint32).sum(axis=1).astype(np.uint32)
            db.execute("SELECT vector_quantize_preload('t','v')")
            got = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
            out["%s/%s/rowids" % (which, name)] = np.array([g[0] for g in got], dtype=np.int64)
            out["%s/%s/dist" % (which, name)] = np.array([g[1] for g in got], dtype=np.float32).view(np.uint32)
            db.close()
    return out


if __name__ == "__main__":
    orc.build(ref=True)
    assert orc.have_ref(), "needs /root/reference (run in the build container)"
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **gen_kernels())
    np.savez_compressed(os.path.join(HERE, "sql.npz"), **gen_sql())
    for f in ("kernels.npz", "sql.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
