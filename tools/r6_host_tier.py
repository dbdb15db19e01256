"""GPU-box tool (round 6): the host-resident tier of an out-of-core table.
 (1) C-ABI: ROWS x 384 f32 in pinned host memory (vg_host_alloc), one query streamed through vg_slab_scan_rows slab by slab: GB/s over the host link
 (2) SQL: a file database of SQLROWS x 384 f32 under VECTORGPU_HBM_LIMIT: first scan (reads the table into the host copy), later scans (stream it),
     the same with VECTORGPU_HOST_LIMIT=0 (round 5: the table is read through sqlite3_step for every scan)"""
import ctypes as C, json, os, sqlite3, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
lib = pkg.lib()
dim, k = 384, 20
rows_n = int(os.environ.get("ROWS", "40000000"))
row_bytes = dim * 4
p = C.c_void_p()
t0 = time.perf_counter()
rc = lib.vg_host_alloc(rows_n * row_bytes, C.byref(p))
print("vg_host_alloc %.1f GB: rc %d in %.2f s" % (rows_n * row_bytes / 1e9, rc, time.perf_counter() - t0), flush=True)
if rc == 0:
    buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(rows_n, dim))
    rng = np.random.default_rng(5)
    blk = 1_000_000
    base = rng.standard_normal((blk, dim), dtype=np.float32)
    for r0 in range(0, rows_n, blk):                        # (the same block with a per-block offset: filling 61 GB from the generator takes minutes)
        n = min(blk, rows_n - r0)
        np.add(base[:n], np.float32(1e-3 * (r0 // blk)), out=buf[r0:r0 + n])
    q = base[7].copy()
    for slab_rows in (4_000_000, 8_000_000):
        for rep in range(2):
            t0 = time.perf_counter()
            s = pkg.SlabScan(pkg.F32, dim, pkg.L2, q, k, slab_rows)
            for r0 in range(0, rows_n, slab_rows):
                n = min(slab_rows, rows_n - r0)
                rcs = lib.vg_slab_scan_rows(s.h, C.c_void_p(p.value + r0 * row_bytes), C.c_int64(n), C.c_int64(row_bytes), None)
                if rcs != 0:
                    print("vg_slab_scan_rows rc", rcs, lib.vg_last_error().decode()); break
            ids, dist = s.finish()
            el = time.perf_counter() - t0
            s.close()
            print("C-ABI: %d x %d f32 from pinned host memory, slabs of %d rows: %.3f s = %.1f GB/s, best row %d distance %.4g" % (rows_n, dim, slab_rows, el, rows_n * row_bytes / el / 1e9, ids[0], dist[0]), flush=True)
    lib.vg_host_free(p)
# ---- through SQL
n_sql = int(os.environ.get("SQLROWS", "4000000"))
tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
path = os.path.join(tmp, "host_tier.db")
db = sqlite3.connect(path, isolation_level=None)
db.execute("PRAGMA journal_mode=OFF"); db.execute("PRAGMA synchronous=OFF")
db.execute("CREATE TABLE t (id INTEGER PRIMARY KEY, v BLOB)")
rng = np.random.default_rng(6)
t0 = time.perf_counter()
db.execute("BEGIN")
for r0 in range(0, n_sql, 200_000):
    blkv = rng.standard_normal((min(200_000, n_sql - r0), dim), dtype=np.float32)
    db.executemany("INSERT INTO t(id, v) VALUES (?, ?)", ((r0 + i + 1, blkv[i].tobytes()) for i in range(blkv.shape[0])))
db.execute("COMMIT")
db.close()
print("SQL: table of %d x %d f32 (%.1f GB) written in %.1f s" % (n_sql, dim, n_sql * row_bytes / 1e9, time.perf_counter() - t0), flush=True)
qv = rng.standard_normal(dim, dtype=np.float32)
os.environ["VECTORGPU_HBM_LIMIT"] = os.environ.get("HBM_LIMIT", "1G")
for host_limit in (None, "0"):
    if host_limit is None: os.environ.pop("VECTORGPU_HOST_LIMIT", None)
    else: os.environ["VECTORGPU_HOST_LIMIT"] = host_limit
    d = sqlite3.connect(path, isolation_level=None)
    d.enable_load_extension(True); d.load_extension(pkg.EXT_PATH[:-3])
    d.execute("SELECT vector_init('t', 'v', 'type=FLOAT32,dimension=%d,distance=L2')" % dim)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        res = d.execute("SELECT rowid, distance FROM vector_full_scan('t', 'v', ?, %d)" % k, (qv.tobytes(),)).fetchall()
        times.append(time.perf_counter() - t0)
    st = json.loads(d.execute("SELECT vector_gpu_stats()").fetchone()[0])
    mem = json.loads(d.execute("SELECT vector_gpu_memory('t','v')").fetchone()[0])["column"]
    print("SQL: VECTORGPU_HBM_LIMIT=%s VECTORGPU_HOST_LIMIT=%s: first scan %.3f s, later scans %s s (%.1f GB/s effective), out_of_core %d, host_resident_bytes %d, host_tier_scans %d fills %d, best %s" % (
        os.environ["VECTORGPU_HBM_LIMIT"], host_limit, times[0], ["%.3f" % x for x in times[1:]], n_sql * row_bytes / min(times[1:]) / 1e9, mem["out_of_core"], mem["host_resident_bytes"],
        st["host_tier_scans"], st["host_tier_fills"], res[0]), flush=True)
    d.close()
os.remove(path)
