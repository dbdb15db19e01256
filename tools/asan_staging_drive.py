"""driven by tools/asan_host_check.sh: the extension's staging code - the single loop, the parallel reader threads, their fallbacks,
vector_quantize's staging in front of its transaction - over the host-memory stub engine (tools/asan_stub_engine.c), under a sanitizer"""
import os, sys, sqlite3, tempfile, json
import numpy as np
ext = sys.argv[1]
n, dim, k = 260_000, 8, 5
rng = np.random.default_rng(1)
rows = rng.standard_normal((n, dim), dtype=np.float32)
ids = np.cumsum(rng.integers(1, 4, n)).astype(np.int64)
q = rows[1234].copy()
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "s.db")
db = sqlite3.connect(path, isolation_level=None)
db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
db.execute("BEGIN")
db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", [(int(ids[i]), None if i % 5000 == 7 else rows[i].tobytes()) for i in range(n)])
db.execute("COMMIT")
db.close()

def connect():
    d = sqlite3.connect(path, isolation_level=None)
    d.enable_load_extension(True)
    d.load_extension(ext)
    d.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
    return d

sql = "SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, %d)" % k
res = {}
for threads in ("1", "4", "8"):
    os.environ["VECTORGPU_STAGE_THREADS"] = threads
    d = connect()
    res[threads] = d.execute(sql, (q.tobytes(),)).fetchall()
    st = json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])
    print("threads", threads, res[threads][0], "parallel passes so far", st["parallel_reader_passes"], json.loads(d.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["total_bytes"] > 0)
    d.execute("INSERT INTO t(id, v) VALUES (?, ?)", (int(ids[-1]) + 9, q.tobytes()))          # append: the watermark path
    assert d.execute(sql, (q.tobytes(),)).fetchall()[0][0] in (int(ids[1234]), int(ids[-1]) + 9)
    d.execute("DELETE FROM t WHERE id = ?", (int(ids[-1]) + 9,))
    d.close()
assert res["1"] == res["4"] == res["8"] and res["1"][0][0] == int(ids[1234])
os.environ["VECTORGPU_STAGE_THREADS"] = "4"
d = connect()                                            # vector_quantize: staged in front of its BEGIN, then a failing option string
print("quantize", d.execute("SELECT vector_quantize('t', 'v')").fetchone())
try: d.execute("SELECT vector_quantize('t', 'v', 'qtype=BOGUS')").fetchone()
except sqlite3.Error as e: print("ERR", e)
d.execute("INSERT INTO t(id, v) VALUES (900000000, ?)", (q.tobytes(),))
print(d.execute(sql, (q.tobytes(),)).fetchall()[:2])
d.execute("DELETE FROM t WHERE id = 900000000")
d.execute("CREATE TEMP TABLE t (id INTEGER PRIMARY KEY, v BLOB)")                           # shadowed by a TEMP table: single loop
d.executemany("INSERT INTO temp.t VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(100)])
d.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
print(d.execute(sql, (rows[3].tobytes(),)).fetchall()[0])
d.close()
d = connect()                                            # held exclusively: the readers' probe fails, single loop
d.execute("PRAGMA locking_mode=EXCLUSIVE")
d.execute("INSERT INTO t(id, v) VALUES (900000001, ?)", (q.tobytes(),))
print(d.execute(sql, (q.tobytes(),)).fetchall()[0])
d.execute("DELETE FROM t WHERE id = 900000001")
d.execute("UPDATE t SET v = x'0011' WHERE id = ?", (int(ids[100000]),))                     # a short BLOB in the middle of a range
d.close()
d = connect()
try: d.execute(sql, (q.tobytes(),)).fetchall()
except sqlite3.Error as e: print("ERR", e)
d.close()
# row-granular freshness (track_changes=1): UPDATE / DELETE / INSERT are patched into the staged copy from the update hook's log - the
# answers must be those of a connection that stages the table afresh
d = connect()
d.execute("UPDATE t SET v = ? WHERE id = ?", (rows[0].tobytes(), int(ids[100000])))                   # (repair the short BLOB)
d.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2,track_changes=1')" % dim)
d.execute(sql, (q.tobytes(),)).fetchall()
s0 = json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])["rows_staged"]
d.execute("UPDATE t SET v = ? WHERE id IN (?, ?, ?)", (q.tobytes(), int(ids[10]), int(ids[150000]), int(ids[-1])))
d.execute("DELETE FROM t WHERE id IN (?, ?)", (int(ids[1234]), int(ids[20])))
d.execute("UPDATE t SET v = NULL WHERE id = ?", (int(ids[30]),))
d.execute("INSERT INTO t(id, v) VALUES (?, ?)", (int(ids[-1]) + 100, q.tobytes()))
got = d.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, 6)", (q.tobytes(),)).fetchall()
s1 = json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])["rows_staged"]
d.close()
os.environ["VECTORGPU_STAGE_THREADS"] = "1"
d2 = connect()
want = d2.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, 6)", (q.tobytes(),)).fetchall()
d2.close()
assert sorted(got[:4]) == sorted(want[:4]) and got[0][1] == 0.0 and int(ids[1234]) not in [g[0] for g in got], (got, want)
print("tracked changes: rows re-sent to the engine", s1 - s0, got[:4])
# a table beyond VECTORGPU_HBM_LIMIT: nothing is staged, every scan feeds the rows to the engine slab by slab (vext_staging.inc: ooc_plan,
# ooc_scan_full; the stream and batch functions; vector_quantize slab by slab) - the answers of the resident table
os.environ["VECTORGPU_HBM_LIMIT"] = "1"                   # 1 MiB < 8.3 MB
os.environ["VECTORGPU_STAGE_THREADS"] = "1"               # the single statement feeds the slabs ...
d = connect()
got_single = d.execute(sql, (q.tobytes(),)).fetchall()
d.close()
os.environ["VECTORGPU_STAGE_THREADS"] = "4"               # ... or four reader connections over key ranges
s_before = json.loads(connect().execute("SELECT vector_gpu_stats()").fetchone()[0])["parallel_reader_passes"]
d = connect()
got_ooc = d.execute(sql, (q.tobytes(),)).fetchall()
assert got_ooc == got_single, (got_ooc, got_single)
assert json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])["parallel_reader_passes"] == s_before + 1
mem = json.loads(d.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
assert mem["column"]["out_of_core"] == 1 and mem["column"]["staged"] == 0, mem
stream = d.execute("SELECT count(*), min(distance) FROM vector_full_scan_stream('t', 'v', ?)", (q.tobytes(),)).fetchone()
batch = d.execute("SELECT query, id FROM vector_full_scan_batch('t', 'v', ?, 3)", (np.stack([q, rows[5]]).tobytes(),)).fetchall()
print("out of core:", got_ooc[:2], "stream rows", stream, "batch", batch[:3], "quantize", d.execute("SELECT vector_quantize('t', 'v', 'max_memory=1MB')").fetchone())
# the host-resident tier of an out-of-core table (round 6): the rows are read ONCE into (here: plain) host memory, later scans stream that
# copy; a write drops it; VECTORGPU_HOST_LIMIT=0 keeps the statement-per-scan path
assert d.execute(sql, (q.tobytes(),)).fetchall() == got_ooc          # (vector_quantize's own writes moved the stamps: this scan reads the table again)
st_a = json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])
assert st_a["host_tier_fills"] >= 1 and st_a["host_tier_scans"] >= 1, st_a
assert json.loads(d.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["host_resident_bytes"] > 8_000_000
for _ in range(3):
    assert d.execute(sql, (q.tobytes(),)).fetchall() == got_ooc
st_b = json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])
assert st_b["host_tier_fills"] == st_a["host_tier_fills"] and st_b["host_tier_scans"] == st_a["host_tier_scans"] + 3 and st_b["rows_staged"] == st_a["rows_staged"], (st_a, st_b)
d.execute("INSERT INTO t(id, v) VALUES (?, ?)", (int(ids[-1]) + 77, q.tobytes()))            # a write: the copy is dropped and read again
g2 = d.execute(sql, (q.tobytes(),)).fetchall()
assert (int(ids[-1]) + 77, 0.0) in g2, g2
st_c = json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])
assert st_c["host_tier_fills"] == st_b["host_tier_fills"] + 1, (st_b, st_c)
d.execute("DELETE FROM t WHERE id = ?", (int(ids[-1]) + 77,))
assert d.execute(sql, (q.tobytes(),)).fetchall() == got_ooc
os.environ["VECTORGPU_HOST_LIMIT"] = "0"
d3 = connect()
st_d = json.loads(d3.execute("SELECT vector_gpu_stats()").fetchone()[0])
assert d3.execute(sql, (q.tobytes(),)).fetchall() == got_ooc
st_e = json.loads(d3.execute("SELECT vector_gpu_stats()").fetchone()[0])
assert st_e["host_tier_scans"] == st_d["host_tier_scans"] and st_e["out_of_core_scans"] == st_d["out_of_core_scans"] + 1, (st_d, st_e)
assert json.loads(d3.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["host_resident_bytes"] == 0
d3.close()
del os.environ["VECTORGPU_HOST_LIMIT"]
print("host-resident tier: fills", st_c["host_tier_fills"], "scans from it", st_c["host_tier_scans"])
del os.environ["VECTORGPU_HBM_LIMIT"]
d.close()
d = connect()
want_res = d.execute(sql, (q.tobytes(),)).fetchall()
assert [g[0] for g in got_ooc] == [w[0] for w in want_res] and stream[0] == d.execute("SELECT count(v) FROM t").fetchone()[0], (got_ooc, want_res, stream)
assert [b[1] for b in batch if b[0] == 0] == [w[0] for w in want_res[:3]], batch
d.close()
# one staged copy per process (vext_shared.inc): 8 threads x 8 connections over one database file take references to ONE copy; a commit
# from one of them moves everybody (lazily, at their next scan) to the copy of the new file state, and the old one is freed with its last
# reference.  Every thread scans while the others attach / detach: the registry, the per-copy scan mutex and the statistics under TSan.
import threading
os.environ["VECTORGPU_STAGE_THREADS"] = "4"
base = json.loads(connect().execute("SELECT vector_gpu_stats()").fetchone()[0])
conns, errors = [[] for _ in range(8)], []
barrier = threading.Barrier(8)
def worker(i):
    try:
        for _ in range(8):
            c = sqlite3.connect(path, isolation_level=None, check_same_thread=False, timeout=60)
            c.enable_load_extension(True)
            c.load_extension(ext)
            c.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
            conns[i].append(c)
        barrier.wait()
        for c in conns[i]:
            assert c.execute(sql, (q.tobytes(),)).fetchall() == want[:k], "shared scan"
        barrier.wait()
        if i == 0:                                        # one writer: a new best row
            conns[0][0].execute("INSERT INTO t(id, v) VALUES (?, ?)", (int(ids[-1]) + 500, q.tobytes()))
        barrier.wait()
        for c in conns[i]:
            got = c.execute(sql, (q.tobytes(),)).fetchall()
            assert got[0][1] == 0.0 and int(ids[-1]) + 500 in [g[0] for g in got], ("after the write", got)
        barrier.wait()
    except Exception as e:                                # noqa
        errors.append((i, repr(e)))
        try: barrier.abort()
        except Exception: pass
want = connect().execute(sql, (q.tobytes(),)).fetchall()
th = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
[t.start() for t in th]; [t.join() for t in th]
assert not errors, errors
st = json.loads(conns[0][0].execute("SELECT vector_gpu_stats()").fetchone()[0])
mem = json.loads(conns[3][5].execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])
print("shared copies", st["shared_copies"], "references", st["shared_references"], "sharers seen by one connection", mem["column"]["sharers"],
      "stage passes for 2 x 64 first scans", st["stage_passes"] - base["stage_passes"])
assert st["shared_copies"] == 1 and st["shared_references"] == 64 and mem["column"]["sharers"] == 64
assert st["stage_passes"] - base["stage_passes"] <= 4       # (the copy of each file state is staged once, not 64 times)
for cs in conns:
    for c in cs: c.close()
# copy-on-write: a writer among holders clones the copy (vg_shards_clone) and appends its own row; the other holder moves over for free
a, b = connect(), connect()
qq = rows[77].copy()
ra = a.execute(sql, (qq.tobytes(),)).fetchall()
assert b.execute(sql, (qq.tobytes(),)).fetchall() == ra and ra[0][0] == int(ids[77])
staged0 = json.loads(a.execute("SELECT vector_gpu_stats()").fetchone()[0])["rows_staged"]
a.execute("INSERT INTO t(id, v) VALUES (?, ?)", (int(ids[-1]) + 600, qq.tobytes()))
ga = a.execute(sql, (qq.tobytes(),)).fetchall()
assert [g[0] for g in ga[:2]] == [int(ids[77]), int(ids[-1]) + 600], ga
assert json.loads(a.execute("SELECT vector_gpu_stats()").fetchone()[0])["rows_staged"] - staged0 == 1
assert b.execute(sql, (qq.tobytes(),)).fetchall() == ga
assert json.loads(b.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]["sharers"] == 2
a.execute("DELETE FROM t WHERE id = ?", (int(ids[-1]) + 600,))
a.close(); b.close()
print("copy-on-write: one row re-sent for a commit among two holders")
# a write to a TEMP table moves sqlite3_total_changes, not the file - the connection must KEEP its shared reference
# (round 5 cloned the whole corpus on the device and never shared it again)
a, b = connect(), connect()
ra = a.execute(sql, (qq.tobytes(),)).fetchall()
assert b.execute(sql, (qq.tobytes(),)).fetchall() == ra
passes0 = json.loads(a.execute("SELECT vector_gpu_stats()").fetchone()[0])["stage_passes"]
a.execute("CREATE TEMP TABLE scratch (x)")
a.execute("INSERT INTO scratch VALUES (1), (2), (3)")
assert a.execute(sql, (qq.tobytes(),)).fetchall() == ra
ma = json.loads(a.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]
sa = json.loads(a.execute("SELECT vector_gpu_stats()").fetchone()[0])
assert ma["sharers"] == 2 and sa["shared_copies"] == 1 and sa["stage_passes"] == passes0, (ma, sa, passes0)
a.close(); b.close()
print("a TEMP-table write keeps the shared copy (sharers 2, no staging pass)")
# the staging pass of one table must not hold up first scans of ANOTHER table (round 5: one process-wide mutex across
# the whole pass).  Eight threads, two tables, every thread its own connection: all answers right, each table staged once per file state.
d = connect()
d.execute("CREATE TABLE u (id INTEGER PRIMARY KEY, v BLOB)")
d.execute("BEGIN")
d.executemany("INSERT INTO u VALUES (?, ?)", [(i + 1, rows[i].tobytes()) for i in range(50_000)])
d.execute("COMMIT")
d.close()
keep = connect()                                         # (the library - and its statistics - live as long as one connection has it loaded)
base2 = json.loads(keep.execute("SELECT vector_gpu_stats()").fetchone()[0])
errors2, bar2 = [], threading.Barrier(8)
def worker2(i):
    try:
        c = sqlite3.connect(path, isolation_level=None, check_same_thread=False, timeout=60)
        c.enable_load_extension(True)
        c.load_extension(ext)
        tbl = "t" if i % 2 == 0 else "u"
        c.execute("SELECT vector_init('%s', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % (tbl, dim))
        bar2.wait()
        got = c.execute("SELECT rowid, distance FROM vector_full_scan('%s', 'v', ?, 1)" % tbl, (rows[9].tobytes(),)).fetchall()
        assert got == [((int(ids[9]) if tbl == "t" else 10), 0.0)], (tbl, got)
        bar2.wait()
        c.close()
    except Exception as e:                                # noqa
        errors2.append((i, repr(e)))
        try: bar2.abort()
        except Exception: pass
th = [threading.Thread(target=worker2, args=(i,)) for i in range(8)]
[t.start() for t in th]; [t.join() for t in th]
assert not errors2, errors2
st2 = json.loads(keep.execute("SELECT vector_gpu_stats()").fetchone()[0])
keep.close()
assert 2 <= st2["stage_passes"] + st2["parallel_reader_passes"] - base2["stage_passes"] - base2["parallel_reader_passes"] <= 3 and st2["shared_published"] - base2["shared_published"] == 2, (base2, st2)   # one pass and one published copy per table
print("two tables staged side by side: passes", st2["stage_passes"] - base2["stage_passes"], "parallel", st2["parallel_reader_passes"] - base2["parallel_reader_passes"], "rows", st2["rows_staged"] - base2["rows_staged"])
d = connect()
d.execute("DROP TABLE u")
d.close()
d = connect()
assert json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])["shared_copies"] == 0      # the last reference freed it
d.execute("DELETE FROM t WHERE id = ?", (int(ids[-1]) + 500,))
d.close()
print("asan run done (staging)")
