import os, sys, sqlite3, numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import datagen as dg, make_golden as mg
ext = sys.argv[1]
def connect():
    db = sqlite3.connect(":memory:", isolation_level=None); db.enable_load_extension(True); db.load_extension(ext); return db
db = connect()
print(db.execute("select vector_version(), vector_backend()").fetchall())
# conversions, JSON parsing (good and bad), option parsing, quantize host path with all qtypes and tiny max_memory
for t in ("f32","f16","bf16","i8","u8"):
    for js in ("[1,2,3]", "[ 1.5 , -2 ,3 ,]", "[]", "[1, x]", "1,2", "[1e400]", "[" + ",".join("7" for _ in range(1000)) + "]"):
        try: db.execute("select length(vector_as_%s(?))" % t, (js,)).fetchall()
        except sqlite3.Error as e: pass
        try: db.execute("select length(vector_as_%s(?, 3))" % t, (js,)).fetchall()
        except sqlite3.Error as e: pass
    try: db.execute("select vector_as_%s(x'00112233', 3)" % t).fetchall()
    except sqlite3.Error: pass
for vt in dg.ALL_TYPES:
    d = connect()
    rows = dg.corpus(vt, 500, 37, 3)
    d.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
    d.executemany("INSERT INTO t VALUES (?,?)", [(i+1, rows[i].tobytes()) for i in range(500)])
    d.execute("INSERT INTO t VALUES (9999, NULL)")
    for opts in ("type=%s,dimension=37" % mg.TYPE_OPT[vt], "type=%s,dimension=37,distance=cosine,,foo=bar, x" % mg.TYPE_OPT[vt], "dimension=abc", "", "type=NOPE,dimension=3"):
        try: d.execute("SELECT vector_init('t','v',?)", (opts,))
        except sqlite3.Error: pass
    d.execute("SELECT vector_init('t','v',?)", ("type=%s,dimension=37,distance=L2" % mg.TYPE_OPT[vt],))
    for q in ("", "qtype=UINT8", "qtype=INT8,max_memory=1KB", "max_memory=64KB", "qtype=BOGUS", "max_memory=0"):
        try: print(vt, q, d.execute("SELECT vector_quantize('t','v',?)", (q,)).fetchall())
        except sqlite3.Error as e: print(vt, q, "ERR", e)
    for fn in ("vector_quantize_memory('t','v')", "vector_quantize_preload('t','v')", "vector_quantize_cleanup('t','v')", "vector_quantize_preload('nope','v')"):
        try: print(d.execute("SELECT " + fn).fetchall())
        except sqlite3.Error as e: print("ERR", e)
    for sql in ("SELECT * FROM vector_full_scan('t','v',x'00',3)", "SELECT * FROM vector_full_scan('t','v','[1,2]',3)", "SELECT * FROM vector_full_scan('zz','v',x'00',3)",
                "SELECT * FROM vector_full_scan_batch('t','v','[[1],[2]]',3)", "SELECT * FROM vector_quantize_scan('t','v',x'00',3)", "SELECT * FROM vector_full_scan_stream('t','v',x'00')"):
        try: d.execute(sql).fetchall()
        except sqlite3.Error as e: pass
    d.close()
print("asan run done")
# change tracking: the update hook runs for every row change of the connection (no corpus is staged without the GPU engine)
d = connect()
d.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)"); d.execute("CREATE TABLE o (x)")
d.execute("SELECT vector_init('t','v','type=FLOAT32,dimension=4,track_changes=1')")
d.execute("SELECT vector_init('t','v','type=FLOAT32,dimension=4,track_changes=0,scan_filter=1')")
d.execute("SELECT vector_init('t','v','type=FLOAT32,dimension=4,track_changes=1')")
d.executemany("INSERT INTO t VALUES (?,?)", [(i + 1, np.zeros(4, np.float32).tobytes()) for i in range(3000)])
d.execute("UPDATE t SET v = NULL WHERE id % 7 = 0"); d.execute("DELETE FROM t WHERE id < 100"); d.execute("INSERT INTO o VALUES (1)"); d.execute("DELETE FROM t")
try: d.execute("SELECT * FROM vector_full_scan('t','v',?,3)", (np.zeros(4, np.float32).tobytes(),)).fetchall()
except sqlite3.Error as e: print("scan without the engine:", e)
d.close()
print("asan run done (tracking)")
