"""SQL-level tests of the drop-in extension host (sqlite-vector_amd/vector.so), driven the way a user drives the
reference: python sqlite3 + load_extension.  CPU part: surface, option parsing, conversions, vector_quantize (host C)
byte-for-byte against the reference extension / golden fixtures.  GPU part (-m gpu): the table-valued functions
against the golden results of the reference's vector_full_scan / vector_quantize_scan."""
import os
import sqlite3
import time
import subprocess
import sys

import numpy as np
import pytest

import datagen as dg

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

TYPE_OPT = mg.TYPE_OPT
DIST_OPT = mg.DIST_OPT


@pytest.fixture(scope="module")
def ext_path():
    import __graft_entry__ as g
    b = g._load_build()
    b.build_gpu_library()
    p = b.build_extension()
    assert p and os.path.exists(p)
    return p[:-3]                 # sqlite appends the platform suffix; basename must be "vector"


def connect(path):
    db = sqlite3.connect(":memory:", isolation_level=None)      # autocommit (vector_quantize issues BEGIN)
    db.enable_load_extension(True)
    db.load_extension(path)
    return db


def load_table(db, rows, vt, metric, rowids=None, extra=""):
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    ids = rowids if rowids is not None else range(1, rows.shape[0] + 1)
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(int(i), rows[j].tobytes()) for j, i in enumerate(ids)])
    db.execute("SELECT vector_init('t', 'v', ?)",
               ("type=%s,dimension=%d,distance=%s%s" % (TYPE_OPT[vt], rows.shape[1], DIST_OPT[metric], extra),))


# ------------------------------------------------------------------------------------------------- CPU

def test_only_the_entry_point_is_exported(ext_path):
    out = subprocess.run(["nm", "-D", "--defined-only", ext_path + ".so"], capture_output=True, text=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert syms == ["sqlite3_vector_init"], syms            # sqlite-vector.h:29 is the reference's only export


def test_surface_matches_reference(ext_path, orc):
    """same function names / arities and the same four eponymous table-valued functions (sqlite-vector.c:2574-2634)"""
    db = connect(ext_path)
    mine = set(db.execute("SELECT name, narg FROM pragma_function_list WHERE name LIKE 'vector_%'").fetchall())
    mods = set(r[0] for r in db.execute("SELECT name FROM pragma_module_list WHERE name LIKE 'vector_%'").fetchall())
    want_mods = {"vector_full_scan", "vector_quantize_scan", "vector_full_scan_stream", "vector_quantize_scan_stream"}
    assert want_mods <= mods
    assert {"vector_full_scan_batch", "vector_quantize_scan_batch"} <= mods      # additions (batched queries)
    want = {("vector_version", 0), ("vector_backend", 0), ("vector_init", 3), ("vector_quantize", 2),
            ("vector_quantize", 3), ("vector_quantize_memory", 2), ("vector_quantize_preload", 2),
            ("vector_quantize_cleanup", 2)} | {("vector_as_%s" % t, n) for t in ("f32", "f16", "bf16", "i8", "u8") for n in (1, 2)}
    assert want <= mine
    ref_path = orc.ref_extension_path("cpu")
    if ref_path:
        rdb = connect(ref_path)
        ref = set(rdb.execute("SELECT name, narg FROM pragma_function_list WHERE name LIKE 'vector_%'").fetchall())
        # table-valued functions also show up as functions in pragma_function_list on some builds: compare scalars
        assert {f for f in ref if not f[0].endswith("_scan") and not f[0].endswith("_stream")} <= mine | {(m, -1) for m in want_mods}
    assert db.execute("SELECT vector_version()").fetchone()[0].startswith("0.9.23")
    assert db.execute("SELECT count(*) FROM _sqliteai_vector").fetchone()[0] == 0


def test_conversions_and_errors_match_reference(ext_path, orc):
    ref_path = orc.ref_extension_path("cpu")
    if not ref_path:
        pytest.skip("reference extension not built")
    db, rdb = connect(ext_path), connect(ref_path)
    for fn in ("f32", "f16", "bf16", "i8", "u8"):
        # (the malformed ones pin what the parser ACCEPTS as much as its messages: blanks, trailing commas, what may follow a number)
        for js in ("[1, 2, 3]", "[0.5,-0.25 , 100]", "[1e-8, 65504, 70000, 3.14159]", "[ 7 ]", "[1,2,3,]", "[]",
                   "[   ", "[", "[ ]", " [1,2", "1,2]", "[1 2]", "[a]", "[1,,2]", "[300]", "[-1]", "[1e400]", "[ 1 , 2 , ]", "[1,2,", "[1,2, ",
                   "[1;2]", "", "[1,2]]", "[nan, inf]", "[1,2 ", "[,1]", "[-129, 5]", "[0x10]", "  [ 7 ]  "):
            try:
                a = db.execute("SELECT vector_as_%s(?)" % fn, (js,)).fetchone()[0]
                ea = None
            except sqlite3.Error as e:
                a, ea = None, str(e)
            try:
                b = rdb.execute("SELECT vector_as_%s(?)" % fn, (js,)).fetchone()[0]
                eb = None
            except sqlite3.Error as e:
                b, eb = None, str(e)
            assert a == b and ea == eb, (fn, js, a, b, ea, eb)
    bad = ["SELECT vector_as_f32('1,2')", "SELECT vector_as_f32('[1,x]')", "SELECT vector_as_f32('[1,2]', 3)",
           "SELECT vector_as_f32(x'00010203', 2)", "SELECT vector_as_f16(x'000102')", "SELECT vector_as_f32(12)",
           "SELECT vector_as_u8('[1,256]')", "SELECT vector_as_i8('[-129]')",
           "SELECT vector_init('nope','v','type=FLOAT32,dimension=3')", "SELECT vector_init('t','nope','dimension=3')",
           "SELECT vector_init('t','j','dimension=3')", "SELECT vector_init('t','v','type=FLOAT99,dimension=3')",
           "SELECT vector_init('t','v','type=FLOAT32')", "SELECT vector_init('t','v','dimension=0')",
           "SELECT vector_init('t','v','dimension=3,distance=CHEBYSHEV')", "SELECT vector_init('t','v', 3)",
           "SELECT vector_quantize_preload('t','w')",
           "SELECT * FROM vector_full_scan('t','w',x'00',1)", "SELECT * FROM vector_quantize_scan('t','v',x'000000000000000000000000',1)",
           # last: the reference leaves its BEGIN open when the option string is bad (no ROLLBACK on that path,
           # sqlite-vector.c:1431-1432); ours rolls back.  Only the message is compared.
           "SELECT vector_quantize('t','v','qtype=INT4')"]
    for d in (db, rdb):
        d.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB, j TEXT)")
        d.execute("SELECT vector_init('t','v','type=FLOAT32,dimension=3')")
    for sql in bad:
        errs = []
        for d in (db, rdb):
            try:
                d.execute(sql).fetchall()
                errs.append(None)
            except sqlite3.Error as e:
                errs.append(str(e))
        assert errs[0] == errs[1], (sql, errs)
    # inconsistent re-init is refused with the reference's message, consistent re-init is a no-op
    for sql in ("SELECT vector_init('t','v','type=FLOAT32,dimension=4')", "SELECT vector_init('t','v','type=INT8,dimension=3')",
                "SELECT vector_init('t','v','type=FLOAT32,dimension=3,normalized=1')", "SELECT vector_init('t','v','type=FLOAT32,dimension=3')"):
        errs = []
        for d in (db, rdb):
            try:
                d.execute(sql).fetchall()
                errs.append(None)
            except sqlite3.Error as e:
                errs.append(str(e))
        assert errs[0] == errs[1], (sql, errs)


OPTION_STRINGS = [
    "type=FLOAT32,dimension=3", "t=FLOAT32,d=3", "TYPE = float16 , DIM = 7", "dimension=3", "type=FLOAT32",
    "dimension=3,type=FLOAT32,junk,=5,foo=bar", "type=FLOAT32,dimension=3,distance=cosine", "type=FLOAT32,dimension=3,dist=L1",
    "type=FLOAT32,dimension=3,distance=", "type=FLOAT32,,dimension=3", "type=FLOAT32,dimension=0x10",
    "type=FLOAT32,dimension=3,max_memory=64KBx", "type=FLOAT32,dimension=3,max_memory=1.5 gb", "type=FLOAT32,dimension=3,normalized=2",
    "type=FLOAT32,dimension=3,n=1", "type=BFLOAT16,dimension=4,qtype=INT8", "type=FLOAT32,dimension=3,qtype=uint4",
    "type=FLOAT32,dimension=3,q=UINT8", "type==FLOAT32,dimension=3", "  type=FLOAT32  ,  dimension = 3  ",
    "type=FLOAT32,dimension=3,typ=INT8", "type=FLOAT32;dimension=3", "dimensionx=3,type=FLOAT32", "type=FLOAT32,dimension=-1",
    "type=FLOAT32,dimension=abc", "type=FLOAT32,dimension=3,=", "type=FLOAT32,dimension=3,x=,y=2"]


def test_option_strings_parse_like_the_reference(ext_path, orc):
    """Which keys the option parser recognises (the reference matches its keys by PREFIX, in a fixed order), what it skips,
    and what it rejects: vector_init with each string on both extensions, then three re-inits whose refusal messages name
    the type / dimension / normalized flag the first call stored (sqlite-vector.c:938-1056, :1377-1401)."""
    ref_path = orc.ref_extension_path("cpu")
    if not ref_path:
        pytest.skip("reference extension not built")

    def probe(path, opts):
        d = connect(path)
        d.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
        out = []
        for o in (opts, "type=FLOAT32,dimension=99991", "type=INT8,dimension=3", "type=FLOAT32,dimension=3,normalized=1"):
            try:
                d.execute("SELECT vector_init('t','v',?)", (o,)).fetchall()
                out.append(None)
            except sqlite3.Error as e:
                out.append(str(e))
                if o is opts:
                    break
        return out

    for opts in OPTION_STRINGS:
        assert probe(ext_path, opts) == probe(ref_path, opts), opts


@pytest.mark.parametrize("case", mg.SQL_QUANT_CASES, ids=[c[0] for c in mg.SQL_QUANT_CASES])
def test_vector_quantize_persists_the_reference_bytes(ext_path, case):
    """vector_quantize (host C in round 1) writes the same shadow table + metadata as the reference (golden)."""
    sql = np.load(os.path.join(HERE, "golden", "sql.npz"))
    name, vt, qopt, n, dim, k, seed, nonneg = case
    rows = dg.corpus(vt, n, dim, seed)
    if nonneg:
        rows = np.abs(rows)
    db = connect(ext_path)
    load_table(db, rows, vt, dg.COSINE)
    cnt = db.execute("SELECT vector_quantize('t','v',?)", ("qtype=%s" % qopt,)).fetchone()[0] if qopt else \
        db.execute("SELECT vector_quantize('t','v')").fetchone()[0]
    assert cnt == n
    meta = dict(db.execute("SELECT key, value FROM _sqliteai_vector WHERE tblname='t'").fetchall())
    want = sql["avx2/%s/qparams" % name]
    assert int(meta["qtype"]) == int(want[0]) and meta["qscale"] == want[1] and meta["qoffset"] == want[2]
    blob = b"".join(r[0] for r in db.execute("SELECT data FROM vector0_t_v ORDER BY rowid1").fetchall())
    rec = np.frombuffer(blob, dtype=np.uint8).reshape(n, 8 + dim)
    assert np.array_equal(rec[:, :8].copy().view("<i8").ravel(), np.arange(1, n + 1))
    assert np.array_equal(rec[:64, 8:], sql["avx2/%s/qhead" % name])
    assert np.array_equal(rec[:, 8:].astype(np.uint32).sum(axis=1).astype(np.uint32), sql["avx2/%s/qrowsum" % name])
    assert db.execute("SELECT vector_quantize_memory('t','v')").fetchone()[0] == n * (8 + dim)
    # chunking by max_memory: same bytes, more chunks
    db.execute("SELECT vector_quantize('t','v',?)", ("max_memory=64KB" + (",qtype=%s" % qopt if qopt else ""),))
    nchunks = db.execute("SELECT count(*) FROM vector0_t_v").fetchone()[0]
    assert nchunks == -(-n // (65536 // (8 + dim)))
    blob2 = b"".join(r[0] for r in db.execute("SELECT data FROM vector0_t_v ORDER BY rowid1").fetchall())
    assert blob2 == blob
    db.execute("SELECT vector_quantize_cleanup('t','v')")
    assert db.execute("SELECT count(*) FROM sqlite_master WHERE name='vector0_t_v'").fetchone()[0] == 0


def test_scan_without_gpu_is_a_loud_sql_error(ext_path):
    import __graft_entry__ as g
    if g.load_package().device_count() > 0:
        pytest.skip("a GPU is present")
    db = connect(ext_path)
    rows = dg.corpus(dg.F32, 10, 8, 1)
    load_table(db, rows, dg.F32, dg.L2)
    with pytest.raises(sqlite3.OperationalError) as ei:
        db.execute("SELECT * FROM vector_full_scan('t','v',?,3)", (rows[0].tobytes(),)).fetchall()
    assert "no HIP device" in str(ei.value)
    assert "no device" in db.execute("SELECT vector_backend()").fetchone()[0]


# ------------------------------------------------------------------------------------------------- GPU

def _check_vs_golden(got, want_ids, want_dist_bits, exact):
    ids = [g[0] for g in got]
    dist = np.array([g[1] for g in got], dtype=np.float32)
    want = want_dist_bits.view(np.float32)
    assert len(ids) == len(want_ids)
    if exact:
        assert dg.same_float_bits(dist, want)
    else:
        assert np.allclose(dist, want, rtol=1e-5, atol=1e-6)
    if exact:
        # integer element types (bit-exact distances): the DEFAULT result order is the reference's own (tie_order=reference,
        # vector_ext.c tie_order_for) - EVERY rowid of the reference extension's output, in its order, ties included
        assert ids == [int(x) for x in want_ids]
        return
    # float columns default to (distance, scan position): rowids must match wherever the distance is unique in the result and
    # below the last distance (a row tying with the LAST one may be left out of the result on either side)
    uniq = [i for i in range(len(want)) if np.sum(want == want[i]) == 1 and want[i] < want[-1]]
    assert [ids[i] for i in uniq] == [int(want_ids[i]) for i in uniq]


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.SQL_SCAN_CASES, ids=[c[0] for c in mg.SQL_SCAN_CASES])
def test_vector_full_scan_vs_reference_golden(ext_path, case):
    name, vt, metric, n, dim, k, seed, low = case
    if vt in (dg.F16, dg.BF16):
        import __graft_entry__ as g
        if not hasattr(g.load_package(), "HALF_TYPES"):
            pytest.skip("f16/bf16 kernels land later in round 1")
    sql = np.load(os.path.join(HERE, "golden", "sql.npz"))
    rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
    q = dg.query(vt, dim, seed + 1, low_entropy=low)
    db = connect(ext_path)
    assert "gfx950" in db.execute("SELECT vector_backend()").fetchone()[0]
    load_table(db, rows, vt, metric)
    got = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    _check_vs_golden(got, sql["avx2/%s/rowids" % name], sql["avx2/%s/dist" % name], exact=vt in (dg.U8, dg.I8))
    # id column == rowid, output already ordered, JSON query == BLOB query
    got2 = db.execute("SELECT id, distance FROM vector_full_scan('t','v',?,?) ORDER BY distance", (q.tobytes(), k)).fetchall()
    assert got2 == got
    if vt == dg.F32:
        js = "[" + ",".join(repr(float(x)) for x in q) + "]"
        got3 = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (js, k)).fetchall()
        assert got3 == got
    # staging is invalidated by writes: delete the best row and it disappears from the next scan
    best = got[0][0]
    db.execute("DELETE FROM t WHERE id=?", (best,))
    got4 = db.execute("SELECT rowid FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    assert best not in [g[0] for g in got4] and len(got4) == k


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.SQL_SCAN_CASES + mg.SQL_QUANT_CASES, ids=["filtered-" + c[0] for c in mg.SQL_SCAN_CASES + mg.SQL_QUANT_CASES])
def test_golden_cases_through_the_filter_scans(ext_path, case, monkeypatch):
    """the same golden cases with the lower-bound filter scans forced on for tables of every size (by default they start at 2^20
    rows): int8 shadow copy for f32 / f16 / bf16 tables, high nibbles for the quantized scans - answers unchanged"""
    monkeypatch.setenv("VG_SCAN_FILTER_MIN_MB", "0")
    monkeypatch.setenv("VG_SCAN_FILTER_N4", "1")
    monkeypatch.setenv("VG_SCAN_FILTER_NO_GUARD", "1")
    if case in mg.SQL_QUANT_CASES:
        test_vector_quantize_scan_vs_reference_golden(ext_path, case)
    else:
        test_vector_full_scan_vs_reference_golden(ext_path, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.SQL_QUANT_CASES, ids=[c[0] for c in mg.SQL_QUANT_CASES])
def test_vector_quantize_scan_vs_reference_golden(ext_path, case):
    name, vt, qopt, n, dim, k, seed, nonneg = case
    sql = np.load(os.path.join(HERE, "golden", "sql.npz"))
    rows = dg.corpus(vt, n, dim, seed)
    if nonneg:
        rows = np.abs(rows)
    q = dg.query(vt, dim, seed + 1)
    db = connect(ext_path)
    load_table(db, rows, vt, dg.COSINE)
    db.execute("SELECT vector_quantize('t','v',?)", ("qtype=%s" % qopt,)) if qopt else db.execute("SELECT vector_quantize('t','v')")
    for preload in (False, True):
        if preload:
            db.execute("SELECT vector_quantize_preload('t','v')")
        got = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
        _check_vs_golden(got, sql["avx2/%s/rowids" % name], sql["avx2/%s/dist" % name], exact=True)
    # re-quantizing while preloaded refreshes the HBM copy (sqlite-vector.c:1471)
    db.execute("SELECT vector_quantize('t','v','qtype=INT8')")
    got = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    assert len(got) == k


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in mg.SQL_SCAN_CASES if c[1] in (dg.U8, dg.I8)] + list(mg.SQL_QUANT_CASES),
                         ids=[c[0] for c in mg.SQL_SCAN_CASES if c[1] in (dg.U8, dg.I8)] + [c[0] for c in mg.SQL_QUANT_CASES])
@pytest.mark.parametrize("how", ("default", "default-shards", "option", "env", "reinit", "shards", "store-mode"))
def test_reference_tie_order_matches_golden_rowids(ext_path, case, how, monkeypatch):
    """the reference's own result order - the DEFAULT for integer element types, or asked for (vector_init option,
    VECTORGPU_TIE_ORDER, a later vector_init call); one device or several shards; through the fused form or (VG_REF_STORE_MODE)
    round 2's store-mode replay: integer distances are bit-exact, so EVERY rowid of the reference's own result must come back
    in its order - ties included (the low-entropy cases tie constantly).  Golden = the reference extension's output."""
    sql = np.load(os.path.join(HERE, "golden", "sql.npz"))
    extra = ",tie_order=reference" if how in ("option", "shards") else ""
    if how in ("shards", "default-shards"):
        extra += ",gpu_devices=0+0+0,gpu_shard_rows=257"
    if how == "store-mode":
        monkeypatch.setenv("VG_REF_STORE_MODE", "1")
    if how == "env":
        monkeypatch.setenv("VECTORGPU_TIE_ORDER", "reference")
    db = connect(ext_path)
    if len(case) == 8 and case in mg.SQL_SCAN_CASES:
        name, vt, metric, n, dim, k, seed, low = case
        rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
        q = dg.query(vt, dim, seed + 1, low_entropy=low)
        load_table(db, rows, vt, metric, extra=extra)
        fn = "vector_full_scan"
    else:
        name, vt, qopt, n, dim, k, seed, nonneg = case
        rows = dg.corpus(vt, n, dim, seed)
        if nonneg:
            rows = np.abs(rows)
        q = dg.query(vt, dim, seed + 1)
        load_table(db, rows, vt, dg.COSINE, extra=extra)
        db.execute("SELECT vector_quantize('t','v',?)", ("qtype=%s" % qopt,)) if qopt else db.execute("SELECT vector_quantize('t','v')")
        db.execute("SELECT vector_quantize_preload('t','v')")
        fn = "vector_quantize_scan"
    if how == "reinit":                                      # staged in (distance, position) order first, then switched
        db.execute("SELECT vector_init('t','v',?)", ("type=%s,dimension=%d,tie_order=position" % (TYPE_OPT[vt], dim),))
        db.execute("SELECT rowid FROM %s('t','v',?,?)" % fn, (q.tobytes(), k)).fetchall()
        db.execute("SELECT vector_init('t','v',?)", ("type=%s,dimension=%d,tie_order=reference" % (TYPE_OPT[vt], dim),))
    got = db.execute("SELECT rowid, distance FROM %s('t','v',?,?)" % fn, (q.tobytes(), k)).fetchall()
    want_ids = sql["avx2/%s/rowids" % name]
    want = sql["avx2/%s/dist" % name].view(np.float32)
    assert [g[0] for g in got] == [int(x) for x in want_ids], (name, how)
    assert dg.same_float_bits(np.array([g[1] for g in got], dtype=np.float32), want)
    # the batch TVF in this mode: the batch kernels with one more list slot, tie queries answered again one by one - same rows
    gb = db.execute("SELECT id, distance FROM %s_batch('t','v',?,?) WHERE query = 1" % fn, ((q.tobytes() * 2), k)).fetchall()
    assert [g[0] for g in gb] == [int(x) for x in want_ids]


@pytest.mark.gpu
def test_incremental_staging_appends_only_the_new_rows(ext_path, orc):
    """row-granular freshness: INSERTs behind the staged rows go to the device as an append of just those rows (the
    reference re-reads the table every scan, sqlite-vector.c:2077-2107; round 1 re-staged the whole table); an UPDATE, a
    DELETE, an INSERT below the key watermark or a write to another table fall back to a full re-stage.  Results always
    equal a fresh connection's."""
    import __graft_entry__ as g
    pkg = g.load_package()
    n, dim = 200_000, 8
    rows = dg.corpus(dg.F32, n + 50, dim, 77)
    q = dg.query(dg.F32, dim, 78)
    db = connect(ext_path)
    load_table(db, rows[:n], dg.F32, dg.L2)
    db.execute("CREATE TABLE other (x)")

    def scan():
        return db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,10)", (q.tobytes(),)).fetchall()

    def expect(m):
        d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, m[1])
        ids, dist, _ = orc.topk_ordered(d, np.asarray(m[0], dtype=np.int64), 10)
        return ids.tolist()

    live_ids, live_rows = list(range(1, n + 1)), rows[:n].copy()
    scan()
    base = pkg.lib().vg_stat_rows_appended()
    assert base >= n
    # 1 row appended -> 1 row staged
    db.execute("INSERT INTO t(id, v) VALUES (?, ?)", (n + 1, q.tobytes()))
    got = scan()
    assert pkg.lib().vg_stat_rows_appended() - base == 1
    assert got[0][0] == n + 1 and got[0][1] == 0.0
    live_ids.append(n + 1); live_rows = np.vstack([live_rows, q[None, :]])
    assert [x[0] for x in got] == expect((live_ids, live_rows))
    # several statements, implicit keys, a NULL vector among them (skipped like in a full scan)
    base = pkg.lib().vg_stat_rows_appended()
    for j in range(5):
        db.execute("INSERT INTO t(v) VALUES (?)", (rows[n + j].tobytes(),))
        live_ids.append(n + 2 + j); live_rows = np.vstack([live_rows, rows[n + j][None, :]])
    db.execute("INSERT INTO t(v) VALUES (NULL)")
    got = scan()
    assert pkg.lib().vg_stat_rows_appended() - base == 5
    assert [x[0] for x in got] == expect((live_ids, live_rows))
    # a write to ANOTHER table: the change counter moves, the counts do not match -> full re-stage, same answer
    base = pkg.lib().vg_stat_rows_appended()
    db.execute("INSERT INTO other VALUES (1)")
    got = scan()
    assert pkg.lib().vg_stat_rows_appended() - base == len(live_ids)
    assert [x[0] for x in got] == expect((live_ids, live_rows))
    # an UPDATE of a staged row, a DELETE, an INSERT below the watermark: full re-stage each time
    db.execute("UPDATE t SET v = ? WHERE id = 7", (rows[n + 10].tobytes(),))
    live_rows[6] = rows[n + 10]
    assert [x[0] for x in scan()] == expect((live_ids, live_rows))
    db.execute("DELETE FROM t WHERE id = ?", (n + 1,))
    keep = [i for i, r in enumerate(live_ids) if r != n + 1]
    live_ids = [live_ids[i] for i in keep]; live_rows = live_rows[keep]
    assert [x[0] for x in scan()] == expect((live_ids, live_rows))
    base = pkg.lib().vg_stat_rows_appended()
    db.execute("INSERT INTO t(id, v) VALUES (?, ?)", (n + 1, rows[n + 11].tobytes()))   # fills the hole below MAX(id)
    pos = live_ids.index(n + 2)
    live_ids.insert(pos, n + 1); live_rows = np.insert(live_rows, pos, rows[n + 11], axis=0)
    got = scan()
    assert pkg.lib().vg_stat_rows_appended() - base == len(live_ids)
    assert [x[0] for x in got] == expect((live_ids, live_rows))
    # inside a transaction: appended, rolled back -> the rows are gone again
    db.execute("BEGIN")
    db.execute("INSERT INTO t(v) VALUES (?)", (q.tobytes(),))
    assert scan()[0][1] == 0.0
    db.execute("ROLLBACK")
    assert [x[0] for x in scan()] == expect((live_ids, live_rows))


@pytest.mark.gpu
def test_tracked_changes_patch_delete_and_append_without_a_restage(ext_path, orc):
    """track_changes=1: the update hook tells WHICH rows an UPDATE / DELETE / INSERT touched; the next scan re-reads only those
    rows and patches / removes / appends them in HBM (the reference re-reads the whole table every scan, sqlite-vector.c:2077-2107).
    A write to another table costs nothing.  Statements the hook cannot account for (DELETE without WHERE: truncate optimisation;
    a key UPDATE; REPLACE removing a row through a UNIQUE conflict; a row inserted below the staged keys) fall back to the re-stage.  Results always equal a scan of the table as it is."""
    import __graft_entry__ as g
    pkg = g.load_package()
    n, dim = 120_000, 8
    rows = dg.corpus(dg.F32, n + 200, dim, 177)
    q = dg.query(dg.F32, dim, 178)
    db = connect(ext_path)
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB, tag TEXT UNIQUE)")
    db.executemany("INSERT INTO t(id, v, tag) VALUES (?, ?, ?)", [(i + 1, rows[i].tobytes(), "r%d" % (i + 1)) for i in range(n)])
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2,track_changes=1')" % dim)
    db.execute("CREATE TABLE other (x)")

    def scan(k=10):
        return db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()

    def expect(k=10):
        cur = db.execute("SELECT id, v FROM t WHERE v IS NOT NULL ORDER BY id").fetchall()
        ids = np.array([r[0] for r in cur], dtype=np.int64)
        m = np.frombuffer(b"".join(r[1] for r in cur), dtype=np.float32).reshape(len(cur), dim)
        d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, m)
        oids, odist, _ = orc.topk_ordered(d, ids, k)
        return oids.tolist()

    appended = pkg.lib().vg_stat_rows_appended

    def check(staged_rows, k=10):
        base = appended()
        got = scan(k)
        assert [x[0] for x in got] == expect(k)
        assert appended() - base == staged_rows, (appended() - base, staged_rows)
        return got

    scan()
    # an unrelated table: nothing is staged at all
    db.execute("INSERT INTO other VALUES (1)")
    check(0)
    # UPDATE of a staged row to the query itself: patched in place
    db.execute("UPDATE t SET v = ? WHERE id = 777", (q.tobytes(),))
    got = check(0)
    assert got[0] == (777, 0.0)
    # DELETE of single rows (the best one among them), UPDATE and INSERT in the same round
    db.execute("DELETE FROM t WHERE id IN (777, 5, 119999)")
    db.execute("UPDATE t SET v = ? WHERE id = 60000", (q.tobytes(),))
    db.execute("INSERT INTO t(id, v, tag) VALUES (?, ?, 'new1')", (n + 1, (q * np.float32(1.0001)).astype(np.float32).tobytes()))
    db.execute("INSERT INTO t(v, tag) VALUES (?, 'new2')", (rows[n + 1].tobytes(),))
    got = check(2)                                                 # only the two new rows travel as an append
    assert got[0][0] == 60000 and got[1][0] == n + 1
    # a vector set to NULL leaves the corpus, a NULL one set to a vector can only be appended when it lies behind every staged key
    db.execute("UPDATE t SET v = NULL WHERE id = 60000")
    got = check(0)
    assert got[0][0] == n + 1
    # a run of deletions + rollback of some of them: the log only says where to look, the table decides
    db.execute("BEGIN")
    db.execute("DELETE FROM t WHERE id BETWEEN 1000 AND 1100")
    db.execute("ROLLBACK")
    db.execute("DELETE FROM t WHERE id BETWEEN 2000 AND 2050")
    check(0)
    # inside an open transaction the patched copy is good for one scan (a ROLLBACK moves no stamp): re-staged afterwards
    db.execute("BEGIN")
    db.execute("UPDATE t SET v = ? WHERE id = 10", (q.tobytes(),))
    assert scan()[0][0] == 10
    db.execute("ROLLBACK")
    assert [x[0] for x in scan()] == expect()
    # what the hook does not see: REPLACE removing another row through the UNIQUE tag; a key UPDATE; DELETE without WHERE
    live = db.execute("SELECT COUNT(v) FROM t").fetchone()[0]
    db.execute("INSERT OR REPLACE INTO t(id, v, tag) VALUES (?, ?, 'r42')", (n + 50, rows[n + 50].tobytes()))   # removes id 42
    base = appended(); got = scan(); assert [x[0] for x in got] == expect()
    assert appended() - base >= live - 1                           # noticed by the COUNT check: re-staged
    db.execute("UPDATE t SET id = ? WHERE id = 43", (n + 60,))
    assert [x[0] for x in scan()] == expect()
    db.execute("INSERT INTO t(id, v, tag) VALUES (43, ?, 'mid')", (q.tobytes(),))       # a new row in the middle of the key order
    got = scan()
    assert got[0] == (43, 0.0) and [x[0] for x in got] == expect()
    db.execute("DELETE FROM t")
    assert scan() == []
    db.executemany("INSERT INTO t(id, v, tag) VALUES (?, ?, ?)", [(i + 1, rows[i].tobytes(), "s%d" % i) for i in range(500)])
    assert [x[0] for x in scan()] == expect()
    db.execute("UPDATE t SET v = ? WHERE id = 100", (q.tobytes(),))
    got = check(0)
    assert got[0] == (100, 0.0)
    db.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", (1, 2, 3))
def test_tracked_changes_fuzz_random_statement_mixes(ext_path, orc, seed):
    """track_changes=1 under random mixes of INSERT (behind / between the keys, NULL vectors), UPDATE (vector, vector -> NULL,
    NULL -> vector, key), DELETE (single, ranges, everything), REPLACE, writes to another table, transactions that commit or roll
    back, scans inside transactions: after every round the scan equals the oracle's answer over the table as it is"""
    rng = np.random.default_rng(900 + seed)
    n0, dim = 20_000, 8
    pool = dg.corpus(dg.F32, 60_000, dim, 910 + seed)
    q = dg.query(dg.F32, dim, 920 + seed)
    db = connect(ext_path)
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB, tag INTEGER UNIQUE)")
    db.executemany("INSERT INTO t(id, v, tag) VALUES (?, ?, ?)", [(2 * i + 2, pool[i].tobytes(), i) for i in range(n0)])   # even keys: room between them
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2,track_changes=1')" % dim)
    db.execute("CREATE TABLE other (x)")
    nxt = [n0]

    def fresh():
        nxt[0] += 1
        return pool[nxt[0] % len(pool)], nxt[0]

    def check(k=15):
        got = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
        cur = db.execute("SELECT id, v FROM t WHERE v IS NOT NULL ORDER BY id").fetchall()
        if not cur:
            assert got == []
            return
        ids = np.array([r[0] for r in cur], dtype=np.int64)
        m = np.frombuffer(b"".join(r[1] for r in cur), dtype=np.float32).reshape(len(cur), dim)
        d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, m)
        oids, odist, _ = orc.topk_ordered(d, ids, k)
        assert [x[0] for x in got] == oids.tolist()
        assert np.allclose([x[1] for x in got], odist, rtol=1e-5, atol=1e-6)

    def some_ids(m):
        r = db.execute("SELECT id FROM t ORDER BY random() LIMIT ?", (m,)).fetchall()
        return [x[0] for x in r]

    check()
    import __graft_entry__ as g
    appended = g.load_package().lib().vg_stat_rows_appended
    base, restages = appended(), 0
    for rnd in range(40):
        before = appended()
        in_txn = rng.random() < 0.3
        if in_txn:
            db.execute("BEGIN")
        for _ in range(int(rng.integers(1, 6))):
            op = rng.choice(["ins_tail", "ins_mid", "ins_null", "upd", "upd_best", "upd_null", "upd_from_null", "upd_key", "del", "del_range",
                             "replace", "other", "del_all"], p=[.14, .05, .05, .16, .08, .06, .06, .03, .14, .08, .05, .09, .01])
            if op == "ins_tail":
                for _j in range(int(rng.integers(1, 40))):
                    v, tg = fresh()
                    db.execute("INSERT INTO t(v, tag) VALUES (?, ?)", (v.tobytes(), tg))
            elif op == "ins_mid":
                v, tg = fresh()
                db.execute("INSERT OR IGNORE INTO t(id, v, tag) VALUES (?, ?, ?)", (int(rng.integers(1, 2 * n0)) | 1, v.tobytes(), tg))
            elif op == "ins_null":
                _, tg = fresh()
                db.execute("INSERT INTO t(v, tag) VALUES (NULL, ?)", (tg,))
            elif op in ("upd", "upd_best", "upd_null", "upd_from_null"):
                for r in some_ids(int(rng.integers(1, 30))):
                    if op == "upd_null":
                        db.execute("UPDATE t SET v = NULL WHERE id = ?", (r,))
                    elif op == "upd_best":
                        db.execute("UPDATE t SET v = ? WHERE id = ?", ((q * np.float32(1 + rng.random() * 1e-3)).astype(np.float32).tobytes(), r))
                    else:
                        db.execute("UPDATE t SET v = ? WHERE id = ?", (fresh()[0].tobytes(), r))
                if op == "upd_from_null":
                    db.execute("UPDATE t SET v = ? WHERE v IS NULL", (fresh()[0].tobytes(),))
            elif op == "upd_key":
                r = some_ids(1)
                if r:
                    db.execute("UPDATE OR IGNORE t SET id = ? WHERE id = ?", (int(rng.integers(1, 10**7)), r[0]))
            elif op == "del":
                for r in some_ids(int(rng.integers(1, 50))):
                    db.execute("DELETE FROM t WHERE id = ?", (r,))
            elif op == "del_range":
                a = int(rng.integers(1, 2 * n0))
                db.execute("DELETE FROM t WHERE id BETWEEN ? AND ?", (a, a + int(rng.integers(1, 400))))
            elif op == "replace":
                r = db.execute("SELECT tag FROM t ORDER BY random() LIMIT 1").fetchone()
                if r:
                    v, _ = fresh()
                    db.execute("INSERT OR REPLACE INTO t(v, tag) VALUES (?, ?)", (v.tobytes(), r[0]))      # removes the row that held the tag
            elif op == "other":
                db.execute("INSERT INTO other VALUES (?)", (int(rng.integers(0, 100)),))
            else:
                db.execute("DELETE FROM t")
                db.executemany("INSERT INTO t(v, tag) VALUES (?, ?)", [(fresh()[0].tobytes(), nxt[0]) for _j in range(300)])
            if in_txn and rng.random() < 0.3:
                check()                                            # a scan inside the open transaction
        if in_txn:
            db.execute("ROLLBACK" if rng.random() < 0.5 else "COMMIT")
        check()
        restages += (appended() - before) > 5000                   # (a round that re-staged the table)
    # most rounds are served by patches / compaction / appends of the touched rows; transactions (a copy made inside one is good for
    # one scan), key updates, REPLACE, rows between the keys and DELETE-everything re-stage
    assert restages < 40                                           # (ORDER BY random() makes the mix differ from run to run)
    print("tracked-changes fuzz seed %d: %d of 40 rounds re-staged" % (seed, restages))
    db.close()


@pytest.mark.gpu
def test_dropped_and_recreated_table_is_restaged(ext_path):
    """DROP TABLE t; CREATE TABLE t ... moves neither data_version nor total_changes: PRAGMA schema_version is stamped too"""
    db = connect(ext_path)
    rows = dg.corpus(dg.F32, 50, 8, 5)
    load_table(db, rows, dg.F32, dg.L2)
    q = rows[3].tobytes()
    assert db.execute("SELECT rowid FROM vector_full_scan('t','v',?,1)", (q,)).fetchall() == [(4,)]
    db.execute("DROP TABLE t")
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    assert db.execute("SELECT rowid FROM vector_full_scan('t','v',?,1)", (q,)).fetchall() == []


@pytest.mark.gpu
def test_stream_cursor_survives_restaging_and_cleanup(ext_path, orc):
    """a stream cursor owns its snapshot (distances AND rowids): re-staging, vector_quantize and cleanup on the table while
    it is being stepped neither crash it nor change what it yields (the reference's cursor reads its own statement)"""
    n, dim = 3000, 16
    rows = dg.corpus(dg.F32, n, dim, 91)
    q = dg.query(dg.F32, dim, 92)
    db = connect(ext_path)
    load_table(db, rows, dg.F32, dg.L2, rowids=[5 * i + 2 for i in range(n)])
    want = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, rows)
    cur = db.execute("SELECT rowid, distance FROM vector_full_scan_stream('t','v',?)", (q.tobytes(),))
    first = cur.fetchmany(100)
    db.execute("DELETE FROM t WHERE id < 2000")                                        # stale stamps ...
    db.execute("SELECT count(*) FROM vector_full_scan('t','v',?,5)", (q.tobytes(),)).fetchall()   # ... re-stage (clear + refill)
    db.execute("SELECT vector_quantize('t','v')")
    cq = db.execute("SELECT rowid, distance FROM vector_quantize_scan_stream('t','v',?)", (q.tobytes(),))
    firstq = cq.fetchmany(10)
    db.execute("SELECT vector_quantize_cleanup('t','v')")                              # destroys the quantized corpus
    rest = cur.fetchall()
    got = first + rest
    assert [g[0] for g in got] == [5 * i + 2 for i in range(n)]
    assert np.allclose([g[1] for g in got], want, rtol=1e-5)
    restq = cq.fetchall()
    assert len(firstq) + len(restq) == db.execute("SELECT count(*) FROM t").fetchone()[0]


@pytest.mark.gpu
def test_stream_modules_emit_every_row_once(ext_path, orc):
    n, dim = 700, 48
    rows = dg.corpus(dg.I8, n, dim, 5)
    q = dg.query(dg.I8, dim, 6)
    db = connect(ext_path)
    load_table(db, rows, dg.I8, dg.L1)
    got = db.execute("SELECT rowid, distance FROM vector_full_scan_stream('t','v',?)", (q.tobytes(),)).fetchall()
    want = orc.scan_distances(orc.AVX2, dg.L1, dg.I8, q, rows)
    assert [g[0] for g in got] == list(range(1, n + 1))
    assert dg.same_float_bits(np.array([g[1] for g in got], dtype=np.float32), want)
    # the documented use: filter + order + limit in SQL over GPU-computed distances
    top = db.execute("SELECT rowid FROM vector_full_scan_stream('t','v',?) WHERE rowid % 2 = 0 ORDER BY distance, rowid LIMIT 5",
                     (q.tobytes(),)).fetchall()
    even = [(float(want[i]), i + 1) for i in range(n) if (i + 1) % 2 == 0]
    assert [t[0] for t in top] == [r for _, r in sorted(even)[:5]]
    db.execute("SELECT vector_quantize('t','v')")
    gotq = db.execute("SELECT count(*), min(distance) FROM vector_quantize_scan_stream('t','v',?)", (q.tobytes(),)).fetchone()
    assert gotq[0] == n


@pytest.mark.gpu
def test_null_vectors_without_rowid_tables_and_k0(ext_path, orc):
    db = connect(ext_path)
    db.execute("CREATE TABLE w (k INTEGER PRIMARY KEY, v BLOB) WITHOUT ROWID")
    rows = dg.corpus(dg.F32, 6, 8, 3)
    for i in range(6):
        db.execute("INSERT INTO w VALUES (?, ?)", (100 - i, rows[i].tobytes() if i != 2 else None))
    db.execute("SELECT vector_init('w','v','type=FLOAT32,dimension=8,distance=L1')")
    q = dg.query(dg.F32, 8, 4)
    got = db.execute("SELECT id, distance FROM vector_full_scan('w','v',?,10)", (q.tobytes(),)).fetchall()
    keep = [i for i in range(6) if i != 2]
    d = orc.scan_distances(orc.AVX2, dg.L1, dg.F32, q, rows[keep])
    want = sorted(zip(d.tolist(), [100 - i for i in keep]))
    assert [g[0] for g in got] == [w[1] for w in want] and len(got) == 5
    assert np.allclose([g[1] for g in got], [w[0] for w in want], rtol=1e-5)
    assert db.execute("SELECT id FROM vector_full_scan('w','v',?,0)", (q.tobytes(),)).fetchall() == []


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.SQL_QUANT_CASES, ids=[c[0] for c in mg.SQL_QUANT_CASES])
def test_vector_quantize_on_gpu_persists_the_reference_bytes(ext_path, case):
    """same check as the CPU test above, but on the GPU box vector_quantize takes the GPU path
    (vg_corpus_minmax / vg_corpus_quantize_rows): the shadow table must still be byte-identical to the reference's."""
    import __graft_entry__ as g
    assert g.load_package().device_count() > 0
    test_vector_quantize_persists_the_reference_bytes(ext_path, case)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [dg.L2, dg.DOT, dg.COSINE])
def test_batch_tvf_equals_one_statement_per_query(ext_path, metric):
    """vector_full_scan_batch: (query, id, distance) rows == running vector_full_scan once per query (which is the
    reference's only way to ask several questions).  All three take the matrix-core pass, whose f32 sums associate
    differently from the per-query kernel (same tolerance as the f32 kernels: 1e-5; L2 survivors are re-evaluated with
    the direct formula, so L2 is held to a pure relative bound)."""
    n, dim, k, nq = 5000, 96, 7, 9
    rows = dg.corpus(dg.F32, n, dim, 21)
    qs = np.stack([dg.query(dg.F32, dim, 100 + i) for i in range(nq)])
    db = connect(ext_path)
    load_table(db, rows, dg.F32, metric)
    got = db.execute("SELECT query, id, distance FROM vector_full_scan_batch('t','v',?,?)", (qs.tobytes(), k)).fetchall()
    assert len(got) == nq * k and [g[0] for g in got] == [i for i in range(nq) for _ in range(k)]
    for i in range(nq):
        one = db.execute("SELECT id, distance FROM vector_full_scan('t','v',?,?)", (qs[i].tobytes(), k)).fetchall()
        mine = [(g[1], g[2]) for g in got if g[0] == i]
        assert [m[0] for m in mine] == [o[0] for o in one]
        assert np.allclose([m[1] for m in mine], [o[1] for o in one], rtol=1e-5, atol=1e-5 if metric != dg.L2 else 0)
    # JSON array of arrays == BLOB batch; the usual SQL on top works (best hit per query)
    js = "[" + ",".join("[" + ",".join(repr(float(x)) for x in q) + "]" for q in qs[:3]) + "]"
    gotj = db.execute("SELECT query, id, distance FROM vector_full_scan_batch('t','v',?,?)", (js, k)).fetchall()
    assert gotj == got[: 3 * k]
    best = db.execute("SELECT query, id, min(distance) FROM vector_full_scan_batch('t','v',?,?) GROUP BY query ORDER BY query",
                      (qs.tobytes(), k)).fetchall()
    assert [(b[0], b[1]) for b in best] == [(i, got[i * k][1]) for i in range(nq)]
    # errors and edges
    with pytest.raises(sqlite3.OperationalError, match="multiple of"):
        db.execute("SELECT * FROM vector_full_scan_batch('t','v',?,?)", (qs.tobytes()[:-1], k)).fetchall()
    assert db.execute("SELECT id FROM vector_full_scan_batch('t','v',?,0)", (qs.tobytes(),)).fetchall() == []
    big = db.execute("SELECT count(*) FROM vector_full_scan_batch('t','v',?,?)", (qs[:2].tobytes(), n + 50)).fetchone()[0]
    assert big == 2 * n


@pytest.mark.gpu
def test_vector_gpu_memory_reports_what_the_device_holds(ext_path):
    """vector_gpu_memory(table, column): the row matrix, the per-row copies derived from it and the working buffers, per staged copy
    (raw column / persisted quantization) - vector_quantize_memory keeps the reference's meaning"""
    import json
    n, dim = 3000, 384
    rows = dg.corpus(dg.F32, n, dim, 5)
    db = connect(ext_path)
    load_table(db, rows, dg.F32, dg.COSINE)
    m0 = json.loads(db.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    assert m0["column"]["staged"] == 0 and m0["total_bytes"] == 0
    db.execute("SELECT rowid FROM vector_full_scan('t','v',?,5)", (rows[7].tobytes(),)).fetchall()
    m1 = json.loads(db.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    assert m1["column"]["staged"] == 1 and m1["column"]["rows_bytes"] >= n * dim * 4 and m1["column"]["working_bytes"] > 0
    assert m1["column"]["derived_bytes"] == 0 and m1["quantized"]["staged"] == 0   # (a small corpus: no shadow copy, norms computed in flight)
    db.execute("SELECT count(*) FROM vector_full_scan_batch('t','v',?,5)", (rows[:9].tobytes(),)).fetchone()
    m1b = json.loads(db.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    assert m1b["column"]["derived_bytes"] >= n * 4                             # the batch kernel's cached row norms
    assert m1b["column"]["working_bytes"] > m1["column"]["working_bytes"]      # ... and its query / candidate buffers
    db.execute("SELECT vector_quantize('t','v')")
    db.execute("SELECT vector_quantize_preload('t','v')")
    m2 = json.loads(db.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    assert m2["quantized"]["staged"] == 1 and m2["quantized"]["rows_bytes"] >= n * dim
    assert m2["total_bytes"] == sum(m2[c][f] for c in ("column", "quantized") for f in ("rows_bytes", "derived_bytes", "working_bytes"))
    assert db.execute("SELECT vector_quantize_memory('t','v')").fetchone()[0] == n * (dim + 8)     # the reference's number: the persisted records
    with pytest.raises(sqlite3.OperationalError, match="vector_init"):
        db.execute("SELECT vector_gpu_memory('nope','v')").fetchone()


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [dg.L2, dg.DOT, dg.COSINE])
def test_batch_tvf_over_1536_dimensional_rows(ext_path, metric):
    """vector_full_scan_batch over 1536-dimensional f32 vectors (rows longer than the half-precision matrix-core kernel's registers
    hold: vg_batch_hl.hip splits the K dimension over a workgroup's wavefronts): the same rows as one vector_full_scan per query -
    including a query of zeros and one with a NaN, which leave the matrix pass and get scans of their own."""
    n, dim, k, nq = 6000, 1536, 8, 40
    rows = dg.corpus(dg.F32, n, dim, 23)
    qs = np.stack([dg.query(dg.F32, dim, 300 + i) for i in range(nq)])
    qs[5] = 0.0
    qs[6, 100] = np.float32(np.nan)
    qs[7] = rows[1234]
    db = connect(ext_path)
    load_table(db, rows, dg.F32, metric)
    got = db.execute("SELECT query, id, distance FROM vector_full_scan_batch('t','v',?,?)", (qs.tobytes(), k)).fetchall()
    assert [g[0] for g in got] == sorted(g[0] for g in got)
    for i in range(nq):
        one = db.execute("SELECT id, distance FROM vector_full_scan('t','v',?,?)", (qs[i].tobytes(), k)).fetchall()
        mine = [(g[1], g[2]) for g in got if g[0] == i]
        assert len(one) == (k if i != 6 else 0)                           # (a NaN query: every distance is NaN, none is kept)
        assert [m[0] for m in mine] == [o[0] for o in one], i
        assert np.allclose([m[1] for m in mine], [o[1] for o in one], rtol=1e-5, atol=1e-5 if metric != dg.L2 else 1e-7, equal_nan=True), i
    if metric != dg.DOT:
        assert [g[1] for g in got if g[0] == 7][0] == 1235


@pytest.mark.gpu
@pytest.mark.parametrize("vt", [dg.F16, dg.BF16])
def test_half_precision_batch_tvf_equals_one_statement_per_query(ext_path, vt):
    """f16 / bf16 tables: the batch takes the matrix-core filter + exact re-evaluation (vg_batch_h.hip) and must give
    what one vector_full_scan statement per query gives (same f64 arithmetic, another summation order)."""
    n, dim, k, nq = 6000, 200, 9, 11
    rows = dg.corpus(vt, n, dim, 41)
    qs = np.stack([dg.query(vt, dim, 300 + i) for i in range(nq)])
    for metric in (dg.L2, dg.COSINE, dg.DOT):
        db = connect(ext_path)
        load_table(db, rows, vt, metric)
        got = db.execute("SELECT query, id, distance FROM vector_full_scan_batch('t','v',?,?)", (qs.tobytes(), k)).fetchall()
        assert len(got) == nq * k
        for i in range(nq):
            one = db.execute("SELECT id, distance FROM vector_full_scan('t','v',?,?)", (qs[i].tobytes(), k)).fetchall()
            mine = [(g[1], g[2]) for g in got if g[0] == i]
            assert [m[0] for m in mine] == [o[0] for o in one]
            assert np.allclose([m[1] for m in mine], [o[1] for o in one], rtol=1e-6, atol=1e-7)
        db.close()


@pytest.mark.gpu
def test_quantized_batch_tvf_equals_one_statement_per_query(ext_path):
    n, dim, k, nq = 3000, 64, 5, 4
    rows = np.abs(dg.corpus(dg.F32, n, dim, 31))
    qs = np.abs(np.stack([dg.query(dg.F32, dim, 200 + i) for i in range(nq)]))
    db = connect(ext_path)
    load_table(db, rows, dg.F32, dg.COSINE)
    db.execute("SELECT vector_quantize('t','v')")
    db.execute("SELECT vector_quantize_preload('t','v')")
    got = db.execute("SELECT query, id, distance FROM vector_quantize_scan_batch('t','v',?,?)", (qs.tobytes(), k)).fetchall()
    for i in range(nq):
        one = db.execute("SELECT id, distance FROM vector_quantize_scan('t','v',?,?)", (qs[i].tobytes(), k)).fetchall()
        assert [(g[1], g[2]) for g in got if g[0] == i] == one


@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.SQL_SCAN_CASES[:4], ids=[c[0] for c in mg.SQL_SCAN_CASES[:4]])
def test_extension_over_several_shards_matches_golden(ext_path, case, monkeypatch):
    """VECTORGPU_DEVICES with more than one entry (here three logical shards on device 0): same golden results,
    same stream output, same quantize bytes as the single-device run."""
    monkeypatch.setenv("VECTORGPU_DEVICES", "0,0,0")
    monkeypatch.setenv("VECTORGPU_SHARD_ROWS", "64")
    name, vt, metric, n, dim, k, seed, low = case
    sql = np.load(os.path.join(HERE, "golden", "sql.npz"))
    rows = dg.corpus(vt, n, dim, seed, low_entropy=low)
    q = dg.query(vt, dim, seed + 1, low_entropy=low)
    db = connect(ext_path)
    load_table(db, rows, vt, metric)
    got = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    _check_vs_golden(got, sql["avx2/%s/rowids" % name], sql["avx2/%s/dist" % name], exact=vt in (dg.U8, dg.I8))
    stream = db.execute("SELECT rowid, distance FROM vector_full_scan_stream('t','v',?)", (q.tobytes(),)).fetchall()
    assert [s[0] for s in stream] == list(range(1, n + 1))
    top = sorted(stream, key=lambda r: (r[1], r[0]))[:k]
    assert [t[0] for t in top] == [g[0] for g in got]
    monkeypatch.delenv("VECTORGPU_DEVICES")
    # the same through the option string (the reference ignores unknown keys, so the database stays portable)
    db2 = connect(ext_path)
    load_table(db2, rows, vt, metric, extra=",gpu_devices=0+0,gpu_shard_rows=50")
    got2 = db2.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    assert got2 == got
    db1 = connect(ext_path)
    load_table(db1, rows, vt, metric)
    for d in (db, db1):
        d.execute("SELECT vector_quantize('t','v')")
    a = db.execute("SELECT rowid1, rowid2, counter, data FROM vector0_t_v ORDER BY rowid1").fetchall()
    b = db1.execute("SELECT rowid1, rowid2, counter, data FROM vector0_t_v ORDER BY rowid1").fetchall()
    assert a == b
    ga = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    gb = db1.execute("SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    assert ga == gb


@pytest.mark.gpu
def test_staged_copy_follows_rollbacks_and_other_connections(ext_path, tmp_path):
    """The HBM copy must always equal what "SELECT pk, col FROM tbl" returns NOW (the reference re-reads the table on
    every scan): rows of a rolled-back transaction disappear again (neither PRAGMA data_version nor
    sqlite3_total_changes() moves on ROLLBACK), and another connection's commit is seen."""
    dim, n = 16, 300
    rows = dg.corpus(dg.F32, n, dim, 41)
    q = dg.query(dg.F32, dim, 42)
    path = str(tmp_path / "shared.db")

    def open_db():
        db = sqlite3.connect(path, isolation_level=None)
        db.enable_load_extension(True)
        db.load_extension(ext_path)
        return db

    a = open_db()
    load_table(a, rows, dg.F32, dg.L2)
    sql = "SELECT rowid FROM vector_full_scan('t','v',?,3)"
    base = [r[0] for r in a.execute(sql, (q.tobytes(),)).fetchall()]
    # an exact copy of the query inside a transaction is the nearest row ... until the transaction is rolled back
    a.execute("BEGIN")
    a.execute("INSERT INTO t(id, v) VALUES (100001, ?)", (q.tobytes(),))
    assert a.execute(sql, (q.tobytes(),)).fetchall()[0][0] == 100001
    a.execute("ROLLBACK")
    assert [r[0] for r in a.execute(sql, (q.tobytes(),)).fetchall()] == base
    # savepoints roll back without leaving the transaction
    a.execute("BEGIN")
    a.execute("SAVEPOINT s1")
    a.execute("INSERT INTO t(id, v) VALUES (100002, ?)", (q.tobytes(),))
    assert a.execute(sql, (q.tobytes(),)).fetchall()[0][0] == 100002
    a.execute("ROLLBACK TO s1")
    assert [r[0] for r in a.execute(sql, (q.tobytes(),)).fetchall()] == base
    a.execute("COMMIT")
    assert [r[0] for r in a.execute(sql, (q.tobytes(),)).fetchall()] == base
    # a second connection commits a closer row: connection A's next scan returns it
    b = open_db()
    b.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
    b.execute("INSERT INTO t(id, v) VALUES (100003, ?)", (q.tobytes(),))
    assert a.execute(sql, (q.tobytes(),)).fetchall()[0][0] == 100003
    b.execute("DELETE FROM t WHERE id = 100003")
    assert [r[0] for r in a.execute(sql, (q.tobytes(),)).fetchall()] == base
    a.close(); b.close()


@pytest.mark.gpu
def test_concurrent_connections_from_threads(ext_path, orc):
    """one connection per thread (SQLite's multi-thread mode), all scanning at once: every connection owns its
    context, corpora and HIP stream; results must equal the single-threaded ones."""
    import threading
    dim, n, k, nthreads, reps = 64, 4000, 10, 6, 30
    data = []
    for t in range(nthreads):
        rows = dg.corpus(dg.F32, n, dim, 300 + t)
        qs = [dg.query(dg.F32, dim, 400 + 10 * t + i) for i in range(4)]
        want = []
        for q in qs:
            d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, rows)
            want.append(orc.topk_ordered(d, None, k)[0].tolist())
        data.append((rows, qs, want))
    errors = []

    def worker(t):
        try:
            rows, qs, want = data[t]
            db = sqlite3.connect(":memory:", isolation_level=None, check_same_thread=False)
            db.enable_load_extension(True)
            db.load_extension(ext_path)
            load_table(db, rows, dg.F32, dg.L2)
            for r in range(reps):
                i = r % len(qs)
                got = [x[0] for x in db.execute("SELECT rowid FROM vector_full_scan('t','v',?,?)", (qs[i].tobytes(), k)).fetchall()]
                if got != want[i]:
                    errors.append((t, r, got, want[i]))
                    return
            db.close()
        except Exception as e:                                   # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not errors, errors[:2]


def test_vector_quantize_without_the_engine_reports_why(ext_path, tmp_path):
    """libvectorgpu.so holds the quantizer too: when it cannot be loaded vector_quantize must fail with the loader's
    message and leave no transaction open (it used to report SQLite's "not an error")."""
    script = tmp_path / "q.py"
    script.write_text(
        "import sqlite3, sys\n"
        "db = sqlite3.connect(':memory:', isolation_level=None)\n"
        "db.enable_load_extension(True)\n"
        "db.load_extension(sys.argv[1])\n"
        "db.execute('CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)')\n"
        "db.execute('INSERT INTO t VALUES (1, zeroblob(16))')\n"
        "db.execute(\"SELECT vector_init('t','v','type=FLOAT32,dimension=4')\")\n"
        "try:\n"
        "    db.execute(\"SELECT vector_quantize('t','v')\")\n"
        "    print('NOERROR')\n"
        "except sqlite3.OperationalError as e:\n"
        "    print('ERR', e)\n"
        "print('INTXN', db.in_transaction)\n")
    env = dict(os.environ, VECTORGPU_LIB=str(tmp_path / "missing.so"))
    out = subprocess.run([sys.executable, str(script), ext_path], capture_output=True, text=True, env=env).stdout
    assert "ERR cannot load the GPU engine" in out and "INTXN False" in out, out


RECALL_SQL = """
WITH
exact_knn AS (
    SELECT e.rowid FROM t AS e JOIN vector_full_scan('t', 'v', ?1, ?2) AS v ON e.rowid = v.rowid
),
approx_knn AS (
    SELECT e.rowid FROM t AS e JOIN vector_quantize_scan('t', 'v', ?1, ?2) AS v ON e.rowid = v.rowid
),
matches AS (SELECT COUNT(*) AS match_count FROM exact_knn WHERE rowid IN (SELECT rowid FROM approx_knn)),
total AS (SELECT COUNT(*) AS total_count FROM exact_knn)
SELECT (SELECT match_count FROM matches), (SELECT total_count FROM total),
       CAST((SELECT match_count FROM matches) AS FLOAT) / CAST((SELECT total_count FROM total) AS FLOAT) AS recall;
"""


@pytest.mark.gpu
def test_the_reference_recall_recipe_runs_unchanged(ext_path, orc):
    """QUANTIZATION.md's recall query (the reference's only documented correctness recipe: the table-valued functions
    JOINed with the base table, full scan vs quantized scan) runs as written and gives the same numbers as the
    reference's own extension on the same table."""
    dim, n, k = 64, 3000, 20
    rng = np.random.default_rng(55)
    centers = rng.standard_normal((30, dim)).astype(np.float32)
    rows = (centers[rng.integers(0, 30, n)] + 0.15 * rng.standard_normal((n, dim))).astype(np.float32)   # clustered, like embeddings
    queries = (centers[:5] + 0.15 * rng.standard_normal((5, dim))).astype(np.float32)
    mine = connect(ext_path)
    load_table(mine, rows, dg.F32, dg.COSINE)
    mine.execute("SELECT vector_quantize('t','v')")
    mine.execute("SELECT vector_quantize_preload('t','v')")
    ref_path = orc.ref_extension_path("avx2")
    ref = None
    if ref_path:
        ref = connect(ref_path)
        load_table(ref, rows, dg.F32, dg.COSINE)
        ref.execute("SELECT vector_quantize('t','v')")
        ref.execute("SELECT vector_quantize_preload('t','v')")
    for q in queries:
        m, tot, recall = mine.execute(RECALL_SQL, (q.tobytes(), k)).fetchone()
        assert tot == k and recall >= 0.8, (m, tot, recall)
        if ref is not None:
            assert (m, tot) == ref.execute(RECALL_SQL, (q.tobytes(), k)).fetchone()[:2]


@pytest.mark.gpu
def test_parallel_staging_readers_equal_the_single_loop(ext_path, orc, tmp_path, monkeypatch):
    """a FILE database outside a transaction is staged by several reader connections over disjoint key ranges, appended in key order
    (vext_staging.inc): the same corpus, hence the same answers, as the single sqlite3_step loop - with gaps in the keys, NULL
    vectors, a very uneven key distribution; a short BLOB is the same error; an open transaction and an in-memory database keep
    the single loop."""
    import json
    n, dim, k = 300_000, 8, 12
    rng = np.random.default_rng(5)
    rows = dg.corpus(dg.F32, n, dim, 91)
    ids = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)                 # gaps
    ids[n // 2:] += 10_000_000                                              # ... and a large hole: most key ranges are empty
    q = dg.query(dg.F32, dim, 92)
    path = str(tmp_path / "par.db")
    db0 = sqlite3.connect(path, isolation_level=None)
    db0.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db0.execute("BEGIN")
    db0.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(int(ids[i]), None if i % 1000 == 7 else rows[i].tobytes()) for i in range(n)])
    db0.execute("COMMIT")
    db0.close()
    live = np.array([i % 1000 != 7 for i in range(n)])
    d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, rows[live])
    want_ids, want_d, _ = orc.topk_ordered(d, ids[live], k)

    def open_db(p):
        db = sqlite3.connect(p, isolation_level=None)
        db.enable_load_extension(True)
        db.load_extension(ext_path)
        db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
        return db

    def stats(db):
        return json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])

    sql = "SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, %d)" % k
    res = {}
    for threads in ("1", "4"):
        monkeypatch.setenv("VECTORGPU_STAGE_THREADS", threads)
        db = open_db(path)
        before = stats(db)
        res[threads] = db.execute(sql, (q.tobytes(),)).fetchall()
        after = stats(db)
        assert after["rows_staged"] - before["rows_staged"] == int(live.sum())
        assert after["parallel_reader_passes"] - before["parallel_reader_passes"] == (1 if threads == "4" else 0)
        # an append behind the watermark taken from the parallel pass goes up as one row
        db.execute("INSERT INTO t(id, v) VALUES (?, ?)", (int(ids[-1]) + 5, q.tobytes()))
        got = db.execute(sql, (q.tobytes(),)).fetchall()
        assert got[0][0] == int(ids[-1]) + 5 and got[0][1] == 0.0
        assert stats(db)["rows_staged"] - after["rows_staged"] == 1
        db.execute("DELETE FROM t WHERE id = ?", (int(ids[-1]) + 5,))
        db.close()
    assert res["1"] == res["4"]
    assert [r[0] for r in res["4"]] == want_ids.tolist() and np.allclose([r[1] for r in res["4"]], want_d, rtol=1e-5)
    # inside a transaction other connections cannot see this one's rows: the single loop
    monkeypatch.setenv("VECTORGPU_STAGE_THREADS", "4")
    db = open_db(path)
    db.execute("BEGIN")
    db.execute("INSERT INTO t(id, v) VALUES (1000000000, ?)", (q.tobytes(),))
    before = stats(db)
    got = db.execute(sql, (q.tobytes(),)).fetchall()
    assert got[0][0] == 1000000000 and stats(db)["parallel_reader_passes"] == before["parallel_reader_passes"]
    db.execute("ROLLBACK")
    db.close()
    # a TEMP table of the same name shadows main's in the reference's statement - and is invisible to the readers: the single loop
    db = open_db(path)
    db.execute("CREATE TEMP TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db.executemany("INSERT INTO temp.t(id, v) VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(300)])
    db.execute("INSERT INTO temp.t(id, v) VALUES (777777, ?)", (q.tobytes(),))
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
    before = stats(db)
    got = db.execute(sql, (q.tobytes(),)).fetchall()
    assert got[0][0] == 777777 and got[0][1] == 0.0 and stats(db)["parallel_reader_passes"] == before["parallel_reader_passes"]
    db.close()
    # a database this connection holds exclusively (locking_mode=EXCLUSIVE after a write): other connections get SQLITE_BUSY - the
    # readers' probe sees that at once and the single loop stages, without an error and without waiting for a timeout
    db = open_db(path)
    db.execute("PRAGMA locking_mode=EXCLUSIVE")
    db.execute("INSERT INTO t(id, v) VALUES (2000000000, ?)", (q.tobytes(),))
    before = stats(db)
    t0 = time.perf_counter()
    got = db.execute(sql, (q.tobytes(),)).fetchall()
    assert time.perf_counter() - t0 < 5.0
    assert got[0][0] == 2000000000 and got[0][1] == 0.0 and stats(db)["parallel_reader_passes"] == before["parallel_reader_passes"]
    db.execute("DELETE FROM t WHERE id = 2000000000")
    db.close()
    # a short BLOB is reported like the single loop reports it
    db = open_db(path)
    db.execute("UPDATE t SET v = x'0011' WHERE id = ?", (int(ids[123456]),))
    db.close()
    db = open_db(path)
    with pytest.raises(sqlite3.Error) as ei:
        db.execute(sql, (q.tobytes(),)).fetchall()
    assert "Invalid vector blob found at rowid %d" % int(ids[123456]) in str(ei.value)
    db.close()


@pytest.mark.gpu
def test_wal_snapshot_is_staged_by_the_single_loop_not_by_reader_connections(ext_path, orc, tmp_path, monkeypatch):
    """a statement that calls vector_full_scan already holds a read transaction.  In WAL mode that is a snapshot - reader
    connections opened by the parallel staging pass would see commits made SINCE (and data_version, frozen while the snapshot is held,
    could not tell).  Here another connection commits a new best row after the statement has taken its snapshot and before the scan
    stages the table (a scalar function in front of the scan does the commit): that statement must not see the row - the reference,
    which reads the table inside the statement's own transaction (sqlite-vector.c:2077), would not - and the next statement must."""
    import json
    monkeypatch.setenv("VECTORGPU_STAGE_THREADS", "4")
    n, dim, k = 250_000, 4, 5
    rows = dg.corpus(dg.F32, n, dim, 93)
    q = dg.query(dg.F32, dim, 94)
    path = str(tmp_path / "wal.db")
    db0 = sqlite3.connect(path, isolation_level=None)
    assert db0.execute("PRAGMA journal_mode=WAL").fetchone()[0].lower() == "wal"
    db0.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db0.execute("BEGIN")
    db0.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(n)])
    db0.execute("COMMIT")
    db = sqlite3.connect(path, isolation_level=None)
    db.enable_load_extension(True)
    db.load_extension(ext_path)
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
    poked = []

    def poke():
        if not poked:
            db0.execute("INSERT INTO t(id, v) VALUES (?, ?)", (n + 1, q.tobytes()))      # distance 0: the new best row, committed by ANOTHER connection
            poked.append(1)
        return 1

    db.create_function("poke", 0, poke)
    # the statement's read transaction starts when it starts (it reads t in the first sub-select); poke() then runs before the scan's xFilter
    got = db.execute("SELECT s.c, f.rowid, f.distance FROM (SELECT COUNT(*) + poke() AS c FROM t) AS s, vector_full_scan('t', 'v', ?, %d) AS f" % k,
                     (q.tobytes(),)).fetchall()
    assert poked and got[0][0] == n + 1                                   # (COUNT(*) of the snapshot + 1: the committed row is not in it)
    assert (n + 1) not in [r[1] for r in got], got
    d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, rows)
    want_ids, want_d, _ = orc.topk_ordered(d, None, k)
    assert [r[1] for r in got] == want_ids.tolist()
    st = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])
    assert st["parallel_reader_passes"] == 0, st                         # WAL + an open read transaction: the single loop
    got2 = db.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, %d)" % k, (q.tobytes(),)).fetchall()
    assert got2[0] == (n + 1, 0.0), got2                                  # the next statement sees the commit (data_version moved: re-staged)
    db.close(); db0.close()


@pytest.mark.gpu
def test_quantize_stages_in_front_of_its_transaction_and_reserves_by_key_span(ext_path, tmp_path, monkeypatch):
    """vector_quantize outside a transaction stages the raw column BEFORE its BEGIN (vext_quantize.inc), where the parallel
    readers can serve it, and the pass inside the transaction accepts that copy: the rows go up once.  The HBM reservation comes
    from the key span (two B-tree descents instead of COUNT(*)'s walk over every leaf) - with deleted keys the span is larger than
    the table, the answers are those of the exact count (VECTORGPU_EXACT_COUNT) and of the single loop."""
    import json
    n, dim, k = 250_000, 16, 10
    rows = dg.corpus(dg.F32, n, dim, 17)
    q = dg.query(dg.F32, dim, 18)
    path = str(tmp_path / "qpre.db")
    db0 = sqlite3.connect(path, isolation_level=None)
    db0.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db0.execute("BEGIN")
    db0.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(i - 1000, rows[i].tobytes()) for i in range(n)])   # (negative keys too)
    db0.execute("DELETE FROM t WHERE id % 7 = 3")
    db0.execute("COMMIT")
    live = int(db0.execute("SELECT COUNT(*) FROM t").fetchone()[0])
    db0.close()

    def run(threads, exact):
        monkeypatch.setenv("VECTORGPU_STAGE_THREADS", threads)
        if exact:
            monkeypatch.setenv("VECTORGPU_EXACT_COUNT", "1")
        else:
            monkeypatch.delenv("VECTORGPU_EXACT_COUNT", raising=False)
        db = sqlite3.connect(path, isolation_level=None)
        db.enable_load_extension(True)
        db.load_extension(ext_path)
        db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
        s0 = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])
        assert db.execute("SELECT vector_quantize('t', 'v')").fetchone()[0] == live
        s1 = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])
        full = db.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, %d)" % k, (q.tobytes(),)).fetchall()
        s2 = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])
        quant = db.execute("SELECT rowid, distance FROM vector_quantize_scan('t', 'v', ?, %d)" % k, (q.tobytes(),)).fetchall()
        db.close()
        return s0, s1, s2, full, quant

    s0, s1, s2, full_p, quant_p = run("4", False)
    assert s1["parallel_reader_passes"] - s0["parallel_reader_passes"] == 1
    assert s1["rows_staged"] - s0["rows_staged"] == live                     # once, in front of the BEGIN
    assert s2["rows_staged"] == s1["rows_staged"]                            # the full scan behind it: the copy is still current
    _, _, _, full_1, quant_1 = run("1", True)
    assert full_p == full_1 and quant_p == quant_1
    # a vector_quantize that fails BEHIND its staging pass (a bad option string) must not leave that copy trusted: a row this connection
    # inserts afterwards moves no data_version - the next scan has to see it
    monkeypatch.setenv("VECTORGPU_STAGE_THREADS", "4")
    db = sqlite3.connect(path, isolation_level=None)
    db.enable_load_extension(True)
    db.load_extension(ext_path)
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
    with pytest.raises(sqlite3.Error, match="quantization type"):
        db.execute("SELECT vector_quantize('t', 'v', 'qtype=BOGUS')").fetchone()
    db.execute("INSERT INTO t(id, v) VALUES (5000000, ?)", (q.tobytes(),))
    got = db.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, 3)", (q.tobytes(),)).fetchall()
    assert got[0] == (5000000, 0.0)
    db.execute("DELETE FROM t WHERE id = 5000000")
    db.close()
    # inside a transaction vector_quantize fails like the reference's (its own BEGIN), with nothing staged in front of it
    monkeypatch.setenv("VECTORGPU_STAGE_THREADS", "4")
    db = sqlite3.connect(path, isolation_level=None)
    db.enable_load_extension(True)
    db.load_extension(ext_path)
    db.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
    db.execute("BEGIN")
    before = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])
    with pytest.raises(sqlite3.Error):
        db.execute("SELECT vector_quantize('t', 'v')").fetchone()
    assert json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])["rows_staged"] == before["rows_staged"]
    assert not db.in_transaction                 # (the failure path ends with ROLLBACK, sqlite-vector.c:1450: the caller's transaction is gone)
    db.close()


# ------------------------------------------------------------------------------------------------ out of core
@pytest.mark.gpu
@pytest.mark.parametrize("case", mg.SQL_SCAN_CASES + mg.SQL_QUANT_CASES, ids=["ooc-" + c[0] for c in mg.SQL_SCAN_CASES + mg.SQL_QUANT_CASES])
def test_golden_cases_out_of_core(ext_path, case, monkeypatch):
    """VECTORGPU_HBM_LIMIT below the table's size: nothing is resident, every scan walks the table again through two slabs
    (vg_slabscan.hip; the reference walks the table for every scan too, sqlite-vector.c:2071-2113) - the golden answers of the
    reference extension, the persisted quantization bytes included (vector_quantize runs slab by slab as well)."""
    monkeypatch.setenv("VECTORGPU_HBM_LIMIT", "16K")
    if case in mg.SQL_QUANT_CASES:
        test_vector_quantize_scan_vs_reference_golden(ext_path, case)
        test_vector_quantize_persists_the_reference_bytes(ext_path, case)
    else:
        test_vector_full_scan_vs_reference_golden(ext_path, case)


@pytest.mark.gpu
def test_out_of_core_tables_report_it_and_come_back_when_they_fit(ext_path, orc, monkeypatch):
    import json
    n, dim, k = 30_000, 48, 25
    rows = dg.corpus(dg.F32, n, dim, 9100)
    rows[20_000:20_030] = rows[7]                                                  # exact duplicates: ties across slabs
    q = rows[7].copy()
    db = connect(ext_path)
    load_table(db, rows, dg.F32, dg.L2)
    want = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
    mem = json.loads(db.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    assert mem["column"]["staged"] == 1 and mem["column"]["out_of_core"] == 0
    stream = db.execute("SELECT rowid, distance FROM vector_full_scan_stream('t','v',?) ", (q.tobytes(),)).fetchall()
    db.execute("INSERT INTO t(id, v) VALUES (?, ?)", (n + 1, rows[0].tobytes()))    # (a write: the next scan re-plans)
    monkeypatch.setenv("VECTORGPU_HBM_LIMIT", "1")                                 # 1 MiB < 5.8 MB
    for order in ("position", "reference"):
        monkeypatch.setenv("VECTORGPU_TIE_ORDER", order)
        db2 = connect(ext_path)
        load_table(db2, rows, dg.F32, dg.L2)
        got = db2.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
        mem = json.loads(db2.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
        assert mem["column"]["staged"] == 0 and mem["column"]["out_of_core"] == 1 and mem["column"]["rows_bytes"] == 0
        assert 64 <= mem["column"]["slab_rows"] < n // 4
        if order == "position":
            assert got == want
        else:
            ref = orc.topk_reference(np.array([d for _, d in stream], dtype=np.float32), np.array([i for i, _ in stream], dtype=np.int64), k)
            assert [g[0] for g in got] == ref[0].tolist() and [g[1] for g in got] == ref[1].tolist()
        # the *_stream function and the batch function walk the slabs too
        got_s = db2.execute("SELECT rowid, distance FROM vector_full_scan_stream('t','v',?)", (q.tobytes(),)).fetchall()
        assert got_s == stream
        if order == "position":
            qs = np.stack([rows[7], rows[99], rows[12345]])
            b = db2.execute("SELECT query, id, distance FROM vector_full_scan_batch('t','v',?,?)", (qs.tobytes(), 5)).fetchall()
            for j in range(3):
                one = db2.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,5)", (qs[j].tobytes(),)).fetchall()
                assert [(r, d) for (qn, r, d) in b if qn == j] == one
        # rows written since are seen by the next scan (there is no copy to go stale)
        db2.execute("DELETE FROM t WHERE id=?", (got[0][0],))
        again = db2.execute("SELECT rowid FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
        assert got[0][0] not in [r[0] for r in again] and len(again) == k
        db2.close()
    stats = json.loads(db.execute("SELECT vector_gpu_stats()").fetchone()[0])
    assert stats["out_of_core_scans"] >= 8 and stats["out_of_core_rows"] >= 8 * (n - 1)
    # the limit lifted: the table is resident again
    monkeypatch.delenv("VECTORGPU_HBM_LIMIT")
    monkeypatch.delenv("VECTORGPU_TIE_ORDER")
    db3 = connect(ext_path)
    load_table(db3, rows, dg.F32, dg.L2)
    assert db3.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall() == want
    mem = json.loads(db3.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    assert mem["column"]["staged"] == 1 and mem["column"]["out_of_core"] == 0


@pytest.mark.gpu
def test_out_of_core_host_resident_tier(ext_path, monkeypatch):
    """round 6: a table beyond the device is read ONCE into pinned host memory (vg_host_alloc) and every later scan streams that copy over
    the host link (vector_gpu_stats: host_tier_fills / host_tier_scans; vector_gpu_memory: host_resident_bytes); a write drops the copy;
    VECTORGPU_HOST_LIMIT below the table (or 0) keeps round 5's statement-per-scan path; the answers are the resident table's either way,
    in both tie orders, for rows whose length is not a multiple of 16 bytes too (the repack path of the pinned append)"""
    import json
    for dim in (48, 50):
        n, k = 30_000, 25
        rows = dg.corpus(dg.F32, n, dim, 9300 + dim)
        rows[20_000:20_030] = rows[7]
        q = rows[7].copy()
        db = connect(ext_path)
        load_table(db, rows, dg.F32, dg.L2)
        want = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
        db.close()
        monkeypatch.setenv("VECTORGPU_HBM_LIMIT", "1")
        for host_limit, tier in ((None, True), ("2", False), ("0", False)):          # 2 MiB < the table's 5.8 MB
            if host_limit is None:
                monkeypatch.delenv("VECTORGPU_HOST_LIMIT", raising=False)
            else:
                monkeypatch.setenv("VECTORGPU_HOST_LIMIT", host_limit)
            db2 = connect(ext_path)
            load_table(db2, rows, dg.F32, dg.L2)
            s0 = json.loads(db2.execute("SELECT vector_gpu_stats()").fetchone()[0])
            for _ in range(3):
                assert db2.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall() == want
            s1 = json.loads(db2.execute("SELECT vector_gpu_stats()").fetchone()[0])
            mem = json.loads(db2.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]
            assert mem["out_of_core"] == 1 and mem["staged"] == 0
            if tier:
                assert s1["host_tier_fills"] - s0["host_tier_fills"] == 1 and s1["host_tier_scans"] - s0["host_tier_scans"] == 3, (s0, s1)
                assert mem["host_resident_bytes"] == n * (dim * 4 + 8)
                assert s1["rows_staged"] - s0["rows_staged"] == n                       # the table was read once for three scans
                db2.execute("UPDATE t SET v = ? WHERE id = ?", (rows[1].tobytes(), want[0][0]))   # a write: the copy is dropped, the next scan sees it
                got = db2.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,?)", (q.tobytes(), k)).fetchall()
                assert want[0][0] not in [g[0] for g in got] and got[0][1] == 0.0
                s2 = json.loads(db2.execute("SELECT vector_gpu_stats()").fetchone()[0])
                assert s2["host_tier_fills"] - s1["host_tier_fills"] == 1
            else:
                assert s1["host_tier_scans"] == s0["host_tier_scans"] and mem["host_resident_bytes"] == 0
                assert s1["rows_staged"] - s0["rows_staged"] == 3 * n
            db2.close()
        monkeypatch.delenv("VECTORGPU_HBM_LIMIT")
        monkeypatch.delenv("VECTORGPU_HOST_LIMIT", raising=False)


# ------------------------------------------------------------------------------------------------ one staged copy per process
@pytest.mark.gpu
def test_connections_of_one_process_share_one_staged_copy(ext_path, orc, tmp_path):
    """8 threads x 8 connections over one database file (vext_shared.inc): ONE copy of the column and ONE copy of the preloaded
    quantization on the device, whatever connection asks; a commit moves every connection - at its next scan - to the copy of the new
    file state; the last reference frees the memory.  WAL databases and connections inside a transaction keep copies of their own."""
    import json
    import threading
    import __graft_entry__ as g
    pkg = g.load_package()
    n, dim, k = 200_000, 64, 10
    rows = dg.corpus(dg.F32, n, dim, 9300)
    q = dg.query(dg.F32, dim, 9301)
    path = str(tmp_path / "shared.db")
    db = sqlite3.connect(path, isolation_level=None)
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db.execute("BEGIN")
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(n)])
    db.execute("COMMIT")
    db.close()

    def connect_file(p=path):
        c = sqlite3.connect(p, isolation_level=None, check_same_thread=False, timeout=60)
        c.enable_load_extension(True)
        c.load_extension(ext_path)
        c.execute("SELECT vector_init('t','v','type=FLOAT32,dimension=%d,distance=L2')" % dim)
        return c

    first = connect_file()
    base = json.loads(first.execute("SELECT vector_gpu_stats()").fetchone()[0])
    free0, _ = pkg.device_memory(0)
    sql = "SELECT rowid, distance FROM vector_full_scan('t','v',?,?)"
    want = first.execute(sql, (q.tobytes(), k)).fetchall()
    d = orc.scan_distances(orc.AVX2, dg.L2, dg.F32, q, rows)
    assert [w[0] for w in want] == orc.topk_ordered(d, None, k)[0].tolist()
    first.execute("SELECT vector_quantize('t','v')")
    first.execute("SELECT vector_quantize_preload('t','v')")
    qsql = "SELECT rowid, distance FROM vector_quantize_scan('t','v',?,?)"
    qwant = first.execute(qsql, (q.tobytes(), k)).fetchall()
    free1, _ = pkg.device_memory(0)
    conns, errors = [[] for _ in range(8)], []
    barrier = threading.Barrier(8)

    def worker(i):
        try:
            for _ in range(8):
                conns[i].append(connect_file())
            barrier.wait()
            for c in conns[i]:
                assert c.execute(sql, (q.tobytes(), k)).fetchall() == want
                c.execute("SELECT vector_quantize_preload('t','v')")
                assert c.execute(qsql, (q.tobytes(), k)).fetchall() == qwant
            barrier.wait()
            if i == 0:
                conns[0][0].execute("INSERT INTO t(id, v) VALUES (?, ?)", (n + 1, q.tobytes()))      # a new best row, committed
            barrier.wait()
            for c in conns[i]:
                got = c.execute(sql, (q.tobytes(), k)).fetchall()
                assert got[0] == (n + 1, 0.0) and got[1:] == want[:k - 1]
            barrier.wait()
        except Exception as e:                                   # noqa: BLE001
            errors.append((i, repr(e)))
            try:
                barrier.abort()
            except Exception:                                    # noqa: BLE001
                pass

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors[:3]
    st = json.loads(first.execute("SELECT vector_gpu_stats()").fetchone()[0])
    mem = json.loads(conns[5][2].execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    # `first` has not scanned since the write: it still holds the copy of the old file state (alone), the 64 others share the new one;
    # the quantized records did not change with the INSERT into t - but the file did, so whoever scans them again moves on
    assert mem["column"]["sharers"] == 64 and mem["column"]["staged"] == 1
    assert st["shared_copies"] >= 2 and st["shared_references"] >= 64 + 64 + 1
    # 2 x 64 first scans of the column + 64 preloads: each file state was staged once
    assert st["stage_passes"] - base["stage_passes"] <= 6, (st, base)
    free2, _ = pkg.device_memory(0)
    table_bytes = n * dim * 4
    assert free0 - free2 < 4 * table_bytes + (512 << 20), (free0, free1, free2)         # not 64 copies (3.3 GB): a few, plus working buffers
    # a connection inside a transaction sees its own uncommitted rows: a copy of its own
    c = conns[1][1]
    c.execute("BEGIN")
    c.execute("DELETE FROM t WHERE id=?", (n + 1,))
    assert c.execute(sql, (q.tobytes(), k)).fetchall() == want
    assert json.loads(c.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["sharers"] == 0
    assert conns[2][2].execute(sql, (q.tobytes(), k)).fetchall()[0] == (n + 1, 0.0)      # (the others: the committed state)
    c.execute("ROLLBACK")
    assert c.execute(sql, (q.tobytes(), k)).fetchall()[0] == (n + 1, 0.0)
    assert json.loads(c.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["sharers"] == 64
    for cs in conns:
        for cc in cs:
            cc.close()
    first.close()
    fresh = connect_file()
    assert json.loads(fresh.execute("SELECT vector_gpu_stats()").fetchone()[0])["shared_copies"] == 0
    fresh.close()
    # copy-on-write: a connection that wrote while others hold the copy clones it on the device and appends its own rows - no pass over the table
    a, b = connect_file(), connect_file()
    ra = a.execute(sql, (q.tobytes(), k)).fetchall()
    assert b.execute(sql, (q.tobytes(), k)).fetchall() == ra
    assert json.loads(b.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["sharers"] == 2
    staged0 = json.loads(a.execute("SELECT vector_gpu_stats()").fetchone()[0])["rows_staged"]
    a.execute("INSERT INTO t(id, v) VALUES (?, ?)", (n + 2, (q * np.float32(1.0000001)).tobytes()))
    got = a.execute(sql, (q.tobytes(), k)).fetchall()
    assert {got[0][0], got[1][0]} == {n + 1, n + 2}
    st_a = json.loads(a.execute("SELECT vector_gpu_stats()").fetchone()[0])
    assert st_a["rows_staged"] - staged0 == 1, (st_a["rows_staged"], staged0)          # the one new row, not the table
    assert b.execute(sql, (q.tobytes(), k)).fetchall() == got                         # b moves to a's copy without reading anything
    assert json.loads(b.execute("SELECT vector_gpu_stats()").fetchone()[0])["rows_staged"] - staged0 == 1
    assert json.loads(b.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["sharers"] == 2
    a.close(); b.close()
    # WAL: the change counter of the main file does not follow commits - every connection keeps its own copy
    wal = str(tmp_path / "wal.db")
    db = sqlite3.connect(wal, isolation_level=None)
    db.execute("PRAGMA journal_mode=WAL")
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(3000)])
    db.close()
    a, b = connect_file(wal), connect_file(wal)
    ra = a.execute(sql, (q.tobytes(), k)).fetchall()
    assert b.execute(sql, (q.tobytes(), k)).fetchall() == ra
    assert json.loads(b.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["sharers"] == 0
    a.execute("INSERT INTO t(id, v) VALUES (99999, ?)", (q.tobytes(),))
    assert b.execute(sql, (q.tobytes(), k)).fetchall()[0] == (99999, 0.0)
    a.close(); b.close()


@pytest.mark.gpu
def test_sparse_keys_do_not_leave_an_oversized_reservation(ext_path, tmp_path):
    """the staging pass reserves device memory from the key span (two B-tree descents instead of a COUNT(*) walk); with sparse keys in a
    file that holds other tables too the span is several times the row count - the excess is handed back once the rows have arrived
    (vg_corpus_trim), before the per-row copies sized by the reservation are made"""
    import json
    n, dim = 20_000, 64
    rows = dg.corpus(dg.F32, n, dim, 9500)
    db = sqlite3.connect(str(tmp_path / "sparse.db"), isolation_level=None)
    db.enable_load_extension(True)
    db.load_extension(ext_path)
    db.execute("CREATE TABLE filler (x BLOB)")
    db.execute("BEGIN")
    db.executemany("INSERT INTO filler VALUES (?)", [(b"\0" * 4000,) for _ in range(6000)])          # 24 MB of other pages
    db.execute("COMMIT")
    load_table(db, rows, dg.F32, dg.L2, rowids=[7 * i + 3 for i in range(n)])                       # span = 7 n
    q = rows[11]
    got = db.execute("SELECT rowid, distance FROM vector_full_scan('t','v',?,5)", (q.tobytes(),)).fetchall()
    assert got[0] == (7 * 11 + 3, 0.0)
    mem = json.loads(db.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
    assert n * dim * 4 <= mem["column"]["rows_bytes"] <= int(1.3 * n * dim * 4) + 1024 * dim * 4, mem
    db.execute("INSERT INTO t(id, v) VALUES (?, ?)", (7 * n + 100, q.tobytes()))                     # appends still extend the copy
    got = db.execute("SELECT rowid FROM vector_full_scan('t','v',?,2)", (q.tobytes(),)).fetchall()
    assert sorted(g[0] for g in got) == [7 * 11 + 3, 7 * n + 100]
    db.close()


@pytest.mark.gpu
def test_out_of_core_scan_through_the_parallel_readers(ext_path, orc, tmp_path, monkeypatch):
    """a FILE table of 260 000 rows beyond VECTORGPU_HBM_LIMIT: the slabs are fed by the staging pass' reader connections over key ranges
    (2.5 x one sqlite3_step loop) - the single statement's answer, rowids and distance bits, both tie orders; a writer committing
    between scans is seen by the next one"""
    import json
    n, dim, k = 260_000, 32, 15
    rows = dg.corpus(dg.F32, n, dim, 9600)
    rows[200_000:200_020] = rows[5]
    q = rows[5].copy()
    path = str(tmp_path / "ooc.db")
    db = sqlite3.connect(path, isolation_level=None)
    db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    db.execute("BEGIN")
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(3 * i + 1, rows[i].tobytes()) for i in range(n)])
    db.execute("COMMIT")
    db.close()
    sql = "SELECT rowid, distance FROM vector_full_scan('t','v',?,?)"

    def connect_file():
        c = sqlite3.connect(path, isolation_level=None)
        c.enable_load_extension(True)
        c.load_extension(ext_path)
        c.execute("SELECT vector_init('t','v','type=FLOAT32,dimension=%d,distance=L2')" % dim)
        return c

    resident = connect_file()
    want = resident.execute(sql, (q.tobytes(), k)).fetchall()
    stream = resident.execute("SELECT rowid, distance FROM vector_full_scan_stream('t','v',?)", (q.tobytes(),)).fetchall()
    resident.close()                                                               # (a live copy of the table would be SHARED by the connections below)
    monkeypatch.setenv("VECTORGPU_HBM_LIMIT", "2")                                 # 2 MiB < 33 MB
    res = {}
    for threads in ("1", "4"):
        monkeypatch.setenv("VECTORGPU_STAGE_THREADS", threads)
        c = connect_file()
        p0 = json.loads(c.execute("SELECT vector_gpu_stats()").fetchone()[0])["parallel_reader_passes"]
        t0 = time.perf_counter()
        res[threads] = c.execute(sql, (q.tobytes(), k)).fetchall()
        dt = time.perf_counter() - t0
        st = json.loads(c.execute("SELECT vector_gpu_stats()").fetchone()[0])
        assert st["parallel_reader_passes"] - p0 == (1 if threads == "4" else 0), (threads, st)
        assert json.loads(c.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["out_of_core"] == 1
        print("out-of-core scan of 260k x 32 f32, %s reader thread(s): %.3f s" % (threads, dt))
        c.close()
    assert res["1"] == want and res["4"] == want
    monkeypatch.setenv("VECTORGPU_TIE_ORDER", "reference")
    c = connect_file()
    ref = c.execute(sql, (q.tobytes(), k)).fetchall()
    oref = orc.topk_reference(np.array([d for _, d in stream], dtype=np.float32), np.array([i for i, _ in stream], dtype=np.int64), k)
    assert [r[0] for r in ref] == oref[0].tolist() and [r[1] for r in ref] == oref[1].tolist()
    other = sqlite3.connect(path, isolation_level=None)
    other.execute("DELETE FROM t WHERE id=?", (ref[0][0],))                         # another connection commits
    other.close()
    again = c.execute(sql, (q.tobytes(), k)).fetchall()
    assert ref[0][0] not in [r[0] for r in again] and len(again) == k
    c.close()
