"""Pin the CPU oracle (oracle/oracle.c) to the reference's own code, bit for bit.

The reference is compiled from /root/reference by oracle/Makefile into oracle/_ref/ (stock flags ->
distance-cpu.c path "CPU"; -mavx2 -> distance-avx2.c path "AVX2").  On a box without /root/reference and
without a travelled oracle/_ref these tests skip and tests/test_golden.py carries the pin instead.
"""
import sqlite3

import numpy as np
import pytest

import datagen as dg

DIMS = (1, 3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 384, 768, 1000)


def _pairs(vtype, dim, seed, n=6):
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n):
        if vtype in (dg.F32, dg.F16, dg.BF16):
            sa = np.float32(10.0 ** rng.integers(-3, 3))
            sb = np.float32(10.0 ** rng.integers(-3, 3))
            a = dg.to_storage(vtype, rng.standard_normal(dim).astype(np.float32) * sa)
            b = dg.to_storage(vtype, rng.standard_normal(dim).astype(np.float32) * sb)
        else:
            a = dg.corpus(vtype, 1, dim, seed * 131 + t)[0]
            b = dg.corpus(vtype, 1, dim, seed * 137 + t + 1000)[0]
        out.append((a, b))
    return out


@pytest.mark.parametrize("which", ["cpu", "avx2"])
@pytest.mark.parametrize("vtype", dg.ALL_TYPES)
def test_kernels_random_bit_exact(orc, ref_cpu, ref_avx2, which, vtype):
    ref = ref_cpu if which == "cpu" else ref_avx2
    be = orc.CPU if which == "cpu" else orc.AVX2
    assert ref.backend_name == ("CPU" if which == "cpu" else "AVX2")
    for dim in DIMS:
        for a, b in _pairs(vtype, dim, 7 * dim + vtype):
            for m in dg.ALL_METRICS:
                r = np.float32(ref.distance(m, vtype, a, b))
                o = np.float32(orc.distance(be, m, vtype, a, b))
                assert dg.same_float_bits(r, o), (which, dg.TYPE_NAMES[vtype], dg.METRIC_NAMES[m], dim, r, o)


@pytest.mark.parametrize("which", ["cpu", "avx2"])
@pytest.mark.parametrize("vtype", dg.ALL_TYPES)
def test_kernels_edge_cases_bit_exact(orc, ref_cpu, ref_avx2, which, vtype):
    """zeros, identical vectors, overflow, NaN / Inf / subnormal lanes in block and tail positions."""
    ref = ref_cpu if which == "cpu" else ref_avx2
    be = orc.CPU if which == "cpu" else orc.AVX2
    for dim in (5, 8, 13, 16, 35, 384):
        _, rows = dg.edge_rows(vtype, dim, 1000 + dim)
        for q in dg.edge_queries(vtype, dim, 2000 + dim):
            for m in dg.ALL_METRICS:
                r = ref.scan(m, vtype, q, rows)
                o = np.array([orc.distance(be, m, vtype, q, rows[i]) for i in range(rows.shape[0])], dtype=np.float32)
                assert dg.same_float_bits(r, o), (which, dg.TYPE_NAMES[vtype], dg.METRIC_NAMES[m], dim,
                                                  np.nonzero(r.view(np.uint32) != o.view(np.uint32)), r, o)


def test_backends_agree_on_integers_when_sums_fit_f32(orc):
    """CPU accumulates u8/i8 L2/dot/L1 in float (distance-cpu.c:484,548) and AVX2 in exact ints: the two
    are identical while partial sums stay below 2^24 (dim <= 256 guarantees it)."""
    for vtype in (dg.U8, dg.I8):
        rows = dg.corpus(vtype, 64, 256, 5)
        q = dg.query(vtype, 256, 6)
        for m in dg.ALL_METRICS:
            a = orc.scan_distances(orc.CPU, m, vtype, q, rows)
            b = orc.scan_distances(orc.AVX2, m, vtype, q, rows)
            assert dg.same_float_bits(a, b)


# ----------------------------------------------------------------------------- SQL level

def _connect(orc, which):
    path = orc.ref_extension_path(which)
    if path is None:
        pytest.skip("reference extension not built")
    db = sqlite3.connect(":memory:", isolation_level=None)   # autocommit: vector_quantize issues BEGIN (:1418)
    db.enable_load_extension(True)
    db.load_extension(path)
    return db


TYPE_OPT = {dg.F32: "FLOAT32", dg.F16: "FLOAT16", dg.BF16: "FLOATB16", dg.U8: "UINT8", dg.I8: "INT8"}
DIST_OPT = {dg.L2: "L2", dg.SQUARED_L2: "SQUARED_L2", dg.COSINE: "COSINE", dg.DOT: "DOT", dg.L1: "L1"}


def _load(db, rows, vtype, metric, rowids=None):
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    ids = rowids if rowids is not None else range(1, rows.shape[0] + 1)
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(int(i), rows[j].tobytes()) for j, i in enumerate(ids)])
    db.execute("SELECT vector_init('t', 'v', ?)",
               ("type=%s,dimension=%d,distance=%s" % (TYPE_OPT[vtype], rows.shape[1], DIST_OPT[metric]),))


@pytest.mark.parametrize("which", ["cpu", "avx2"])
@pytest.mark.parametrize("vtype,metric", [(dg.F32, dg.L2), (dg.F32, dg.COSINE), (dg.F32, dg.DOT), (dg.F16, dg.L2),
                                          (dg.BF16, dg.L1), (dg.U8, dg.COSINE), (dg.I8, dg.SQUARED_L2),
                                          (dg.U8, dg.L1)])
def test_full_scan_sql_matches_oracle_topk(orc, which, vtype, metric):
    """vector_full_scan through the reference extension == oracle scan + orc_topk_reference (the slot algorithm,
    sqlite-vector.c:2022-2113), rowids and distances bit for bit — including tie behaviour (low-entropy ints)."""
    be = orc.CPU if which == "cpu" else orc.AVX2
    db = _connect(orc, which)
    n, dim, k = 700, 48, 20
    rows = dg.corpus(vtype, n, dim, 11, low_entropy=(vtype in (dg.U8, dg.I8)))
    q = dg.query(vtype, dim, 12, low_entropy=(vtype in (dg.U8, dg.I8)))
    rowids = np.arange(1, n + 1, dtype=np.int64) * 3 + 5
    _load(db, rows, vtype, metric, rowids)
    got = db.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, ?)", (q.tobytes(), k)).fetchall()
    d = orc.scan_distances(be, metric, vtype, q, rows)
    ids, dist = orc.topk_reference(d, rowids, k)
    assert [g[0] for g in got] == ids.tolist()
    assert dg.same_float_bits(np.array([g[1] for g in got], dtype=np.float32), dist.astype(np.float32))
    # the order-independent contract: same distance sequence as the (distance, position) total order
    ids2, dist2, _ = orc.topk_ordered(d, rowids, k)
    assert dg.same_float_bits(dist2.astype(np.float32), dist.astype(np.float32))


@pytest.mark.parametrize("which", ["cpu", "avx2"])
def test_full_scan_sql_short_table_null_rows_and_k0(orc, which):
    """fewer than k rows -> fewer rows back (sqlite-vector.c:1816-1817); NULL vectors skipped (:2093); k=0 -> empty (:1796)."""
    db = _connect(orc, which)
    rows = dg.corpus(dg.F32, 5, 8, 3)
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    for i in range(5):
        db.execute("INSERT INTO t VALUES (?, ?)", (i + 1, rows[i].tobytes() if i != 2 else None))
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=8')")
    q = dg.query(dg.F32, 8, 4)
    got = db.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, 10)", (q.tobytes(),)).fetchall()
    keep = [0, 1, 3, 4]
    d = orc.scan_distances(orc.CPU if which == "cpu" else orc.AVX2, dg.L2, dg.F32, q, rows[keep])
    ids, dist = orc.topk_reference(d, np.array(keep, dtype=np.int64) + 1, 10)
    assert len(got) == 4 and [g[0] for g in got] == ids.tolist()
    # k == 0: xFilter returns SQLITE_DONE (:1796), which ends the WHOLE statement without a row (even count(*))
    assert db.execute("SELECT rowid FROM vector_full_scan('t', 'v', ?, 0)", (q.tobytes(),)).fetchall() == []
    assert db.execute("SELECT count(*) FROM vector_full_scan('t', 'v', ?, 0)", (q.tobytes(),)).fetchone() is None


@pytest.mark.parametrize("which", ["cpu", "avx2"])
@pytest.mark.parametrize("src_type,qtype_opt", [(dg.F32, None), (dg.F32, "UINT8"), (dg.F16, None), (dg.BF16, "INT8"),
                                                (dg.U8, None), (dg.I8, None)])
def test_quantize_sql_matches_oracle(orc, which, src_type, qtype_opt):
    """vector_quantize -> shadow-table bytes == oracle quant params + quantizer (sqlite-vector.c:1210-1321);
    vector_quantize_scan (preloaded and from disk) == oracle int kernels + slot top-k."""
    be = orc.CPU if which == "cpu" else orc.AVX2
    db = _connect(orc, which)
    n, dim, k = 300, 40, 12
    rows = dg.corpus(src_type, n, dim, 21)
    if src_type == dg.F32 and qtype_opt is None:
        rows = np.abs(rows)                       # non-negative -> AUTO picks U8 (:1258-1261)
    q = dg.query(src_type, dim, 22)
    _load(db, rows, src_type, dg.COSINE)
    if qtype_opt:
        cnt = db.execute("SELECT vector_quantize('t', 'v', ?)", ("qtype=%s" % qtype_opt,)).fetchone()[0]
    else:
        cnt = db.execute("SELECT vector_quantize('t', 'v')").fetchone()[0]
    assert cnt == n
    qt0 = {None: 0, "UINT8": orc.QUANT_U8, "INT8": orc.QUANT_S8}[qtype_opt]
    qt, scale, offset = orc.quant_params(src_type, rows, qt0)
    meta = dict(db.execute("SELECT key, value FROM _sqliteai_vector WHERE tblname='t' AND colname='v'").fetchall())
    assert int(meta["qtype"]) == qt
    assert np.float32(meta["qscale"]) == np.float32(scale) and np.float32(meta["qoffset"]) == np.float32(offset)
    blob = b"".join(r[0] for r in db.execute("SELECT data FROM vector0_t_v ORDER BY rowid1").fetchall())
    rec = np.frombuffer(blob, dtype=np.uint8).reshape(n, 8 + dim)
    assert np.array_equal(rec[:, :8].copy().view("<i8").ravel(), np.arange(1, n + 1))
    qrows = np.stack([orc.quantize(src_type, rows[i], offset, scale, qt) for i in range(n)])
    assert np.array_equal(rec[:, 8:], qrows.view(np.uint8))
    # scan: disk path, then preloaded path
    qq = orc.quantize(src_type, q, offset, scale, qt)
    vt = dg.U8 if qt == orc.QUANT_U8 else dg.I8
    d = orc.scan_distances(be, dg.COSINE, vt, qq, qrows)
    ids, dist = orc.topk_reference(d, None, k)
    for preload in (False, True):
        if preload:
            db.execute("SELECT vector_quantize_preload('t', 'v')")
        got = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t', 'v', ?, ?)", (q.tobytes(), k)).fetchall()
        assert [g[0] for g in got] == ids.tolist(), preload
        assert dg.same_float_bits(np.array([g[1] for g in got], dtype=np.float32), dist.astype(np.float32))


def test_topk_reference_tie_probes(orc):
    """the survey's probes of the history-dependent slot algorithm ('hard parts')."""
    ids, d = orc.topk_reference(np.array([5, 5, 3], dtype=np.float32), None, 2)
    assert sorted(ids.tolist()) == [2, 3] and d.tolist() == [3.0, 5.0]
    ids, d = orc.topk_reference(np.array([3, 5, 5, 5, 1], dtype=np.float32), None, 3)
    assert ids.tolist() == [5, 1, 3]
    # NaN and +Inf never enter; -Inf does
    ids, d = orc.topk_reference(np.array([np.nan, np.inf, -np.inf, 2.0], dtype=np.float32), None, 4)
    assert ids.tolist() == [3, 4] and d.tolist() == [-np.inf, 2.0]
    ids2, d2, pos = orc.topk_ordered(np.array([np.nan, np.inf, -np.inf, 2.0], dtype=np.float32), None, 4)
    assert ids2.tolist() == [3, 4] and pos.tolist() == [2, 3]
