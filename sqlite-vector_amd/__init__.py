"""sqlite-vector_amd: MI355X-native brute-force distance scan + top-k behind sqlite-vector's SQL surface.

This Python module is plumbing only: it builds and loads the two native artefacts and exposes a thin ctypes
view of the C-ABI (include/vectorgpu.h) for tests and bench.py.  The product is

    libvectorgpu.so   HIP kernels + C-ABI                     (csrc/*.hip, hipcc --offload-arch=gfx950)
    vector.so         the SQLite loadable extension, plain C  (ext/vector_ext.c; exports sqlite3_vector_init)

Nothing here computes a distance: without the HIP library / a GPU every call raises.  The package never imports
anything from oracle/ (that is test infrastructure).

The directory name contains a '-', so load it with importlib (see __graft_entry__.load_package) or put the repo
root on sys.path and use importlib.import_module("sqlite-vector_amd").
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VG_LIB_PATH") or os.path.join(HERE, "libvectorgpu.so")   # override: A/B kernel builds
EXT_PATH = os.path.join(HERE, "vector.so")         # must be named vector.* (entry point sqlite3_vector_init)

# enums (same numbering as the reference, distance-cpu.h:36-58)
F32, F16, BF16, U8, I8 = 1, 2, 3, 4, 5
L2, SQUARED_L2, COSINE, DOT, L1 = 1, 2, 3, 4, 5
QUANT_U8, QUANT_S8 = 1, 2
TIE_POSITION, TIE_REFERENCE = 0, 1
KEY_EMPTY = 0xFFFFFFFFFFFFFFFF
HALF_TYPES = True            # f16 / bf16 scan kernels are built in
TYPE_SIZE = {F32: 4, F16: 2, BF16: 2, U8: 1, I8: 1}


class VectorGpuError(RuntimeError):
    pass


_lib = None


# The engine reads its VG_* environment switches when the library is loaded and whenever a corpus / shard set is created - not per query
# (csrc/vg_switches.h).  Tests and tools flip switches between two calls on one corpus: with AUTO_RELOAD every call through this module
# re-reads them first (a few microseconds); bench.py turns it off and calls reload_switches() where it changes the environment.
AUTO_RELOAD = True


def reload_switches():
    _load().vg_reload_switches()


def lib():
    L = _load()
    if AUTO_RELOAD:
        L.vg_reload_switches()
    return L


def _load():
    """Load libvectorgpu.so (fails loudly if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VectorGpuError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    sig = {
        "vg_device_count": (i32, []),
        "vg_backend_name": (C.c_char_p, []),
        "vg_last_error": (C.c_char_p, []),
        "vg_corpus_create": (i32, [i32, i32, i32, i64, C.POINTER(vp)]),
        "vg_corpus_destroy": (None, [vp]),
        "vg_corpus_clear": (i32, [vp]),
        "vg_corpus_reserve": (i32, [vp, i64]),
        "vg_corpus_trim": (i32, [vp]),
        "vg_corpus_clone": (i32, [vp, vp]),
        "vg_shards_clone": (i32, [vp, vp]),
        "vg_shards_trim": (i32, [vp]),
        "vg_corpus_rows": (i64, [vp]),
        "vg_corpus_dim": (i32, [vp]),
        "vg_corpus_type": (i32, [vp]),
        "vg_corpus_device": (i32, [vp]),
        "vg_corpus_hbm_bytes": (i64, [vp]),
        "vg_corpus_set_rowid_base": (i32, [vp, i64]),
        "vg_corpus_append": (i32, [vp, vp, i64, i64, vp]),
        "vg_corpus_append_records": (i32, [vp, vp, i64]),
        "vg_corpus_append_device": (i32, [vp, vp, i64, i64, vp]),
        "vg_scan_topk": (i32, [vp, i32, vp, i32, vp, vp, C.POINTER(i32)]),
        "vg_scan_topk_device": (i32, [vp, i32, vp, i32, vp, vp]),
        "vg_key_distance": (C.c_float, [C.c_uint64]),
        "vg_key_position": (C.c_uint32, [C.c_uint64]),
        "vg_merge_keys": (i32, [vp, i32, i32, vp, i32, vp, vp]),
        "vg_scan_topk_keys": (i32, [vp, i32, vp, i32, vp, C.POINTER(i32)]),
        "vg_scan_topk_enqueue": (i32, [vp, i32, vp, i32]),
        "vg_scan_topk_collect": (i32, [vp, vp]),
        "vg_scan_topk_batch_keys": (i32, [vp, i32, vp, i32, i32, vp, vp]),
        "vg_shards_create": (i32, [vp, i32, i32, i32, i64, C.POINTER(vp)]),
        "vg_shards_destroy": (None, [vp]),
        "vg_shards_clear": (i32, [vp]),
        "vg_shards_count": (i32, [vp]),
        "vg_shards_rows": (i64, [vp]),
        "vg_shards_shard": (vp, [vp, i32]),
        "vg_shards_reserve": (i32, [vp, i64]),
        "vg_shards_set_rowid_base": (i32, [vp, i64]),
        "vg_shards_append": (i32, [vp, vp, i64, i64, vp]),
        "vg_shards_append_records": (i32, [vp, vp, i64]),
        "vg_shards_rowid_at": (i64, [vp, i64]),
        "vg_shards_scan_topk": (i32, [vp, i32, vp, i32, vp, vp, C.POINTER(i32)]),
        "vg_shards_scan_topk_batch": (i32, [vp, i32, vp, i32, i32, vp, vp, vp]),
        "vg_shards_scan_distances": (i32, [vp, i32, vp, vp]),
        "vg_shards_minmax": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32)]),
        "vg_shards_quantize_rows": (i32, [vp, C.c_float, C.c_float, i32, i64, i64, vp]),
        "vg_merge_keys_batch": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, vp]),
        "vg_scan_distances": (i32, [vp, i32, vp, vp]),
        "vg_scan_distances_device": (i32, [vp, i32, vp, vp, vp]),
        "vg_corpus_rowid_at": (i64, [vp, i64]),
        "vg_scan_topk_batch": (i32, [vp, i32, vp, i32, i32, vp, vp, vp]),
        "vg_quantize_query": (i32, [i32, vp, i32, C.c_float, C.c_float, i32, vp]),
        "vg_corpus_minmax": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32)]),
        "vg_corpus_quantize_rows": (i32, [vp, C.c_float, C.c_float, i32, i64, i64, vp]),
        "vg_set_profiling": (i32, [vp, i32]),
        "vg_last_kernel_ms": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "vg_profile_mean_ms": (i32, [vp, C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "vg_scan_kernel_name": (C.c_char_p, [vp, i32]),
        "vg_plan_scan_shape": (i32, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "vg_batch_h_plan": (i32, [i64, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
        "vg_profile_mean_ms_ex": (i32, [vp, C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "vg_corpus_find_rowid": (i64, [vp, i64]),
        "vg_corpus_patch_rows": (i32, [vp, vp, i64, vp, i64]),
        "vg_corpus_delete_rows": (i32, [vp, vp, i64]),
        "vg_shards_find_rowid": (i64, [vp, i64]),
        "vg_shards_patch_rows": (i32, [vp, vp, i64, vp, i64]),
        "vg_shards_delete_rows": (i32, [vp, vp, i64]),
        "vg_filter_exact_evals": (i32, [vp, C.POINTER(C.c_ulonglong)]),
        "vg_filter_guard_cooldown": (i32, [vp]),
        "vg_batch_filter_exact_evals": (i32, [vp, C.POINTER(C.c_ulonglong)]),
        "vg_reload_switches": (None, []),
        "vg_batch_last_path": (i32, [vp]),
        "vg_batch_q8_status": (i32, [vp]),
        "vg_corpus_set_scan_filter": (i32, [vp, i32]),
        "vg_corpus_set_tie_order": (i32, [vp, i32]),
        "vg_corpus_tie_order": (i32, [vp]),
        "vg_corpus_tie_stats": (i32, [vp, vp]),
        "vg_corpus_pass_ms": (i32, [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_longlong)]),
        "vg_shards_set_tie_order": (i32, [vp, i32]),
        "vg_shards_set_scan_filter": (i32, [vp, i32]),
        "vg_shards_set_gather": (i32, [vp, i32]),
        "vg_shards_gather_stats": (i32, [vp, vp]),
        "vg_shards_tie_stats": (i32, [vp, vp]),
        "vg_shards_threaded": (i32, [vp]),
        "vg_slab_scan_begin": (i32, [i32, i32, i32, i32, vp, i32, i32, i64, i64, vp]),
        "vg_slab_scan_rows": (i32, [vp, vp, i64, i64, vp]),
        "vg_slab_scan_records": (i32, [vp, vp, i64]),
        "vg_slab_scan_finish": (i32, [vp, vp, vp, vp]),
        "vg_slab_scan_all": (i32, [vp, vp, vp, vp]),
        "vg_slab_scan_destroy": (None, [vp]),
        "vg_device_memory": (i32, [i32, vp, vp]),
        "vg_host_alloc": (i32, [C.c_size_t, C.POINTER(C.c_void_p)]),
        "vg_host_free": (None, [vp]),
        "vg_corpus_device_bytes": (i32, [vp, vp]),
        "vg_shards_device_bytes": (i32, [vp, vp]),
        "vg_shards_rowids": (i32, [vp, i64, i64, vp]),
        "vg_scan_topk_reference": (i32, [vp, i32, vp, i32, vp, vp, C.POINTER(i32)]),
        "vg_stat_rows_appended": (C.c_longlong, []),
        "vg_reference_topk_replay": (i32, [vp, i64, i32, i64, vp, vp]),
        "vg_reference_topk_replay_slabs": (i32, [vp, i64, i32, i64, i64, vp, vp]),
        "vg_scan_distances_resident": (i32, [vp, i32, vp]),
        "vg_resident_distances_fetch": (i32, [vp, i64, i64, vp]),
        "vg_resident_distances_below": (i32, [vp, i64, C.c_float, vp, i64, C.POINTER(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._sig = sig
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise VectorGpuError("vectorgpu error %d: %s" % (rc, lib().vg_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().vg_device_count()


def backend_name():
    return lib().vg_backend_name().decode()


def plan_batch_half_form(stride_bytes, k, nq):
    """(wavefronts per workgroup, workgroups per CU) of an f16 / bf16 / f32-via-bf16 batch, or None if the matrix-core kernel does not
    serve such rows - host logic only"""
    w, b = C.c_int(0), C.c_int(0)
    if lib().vg_batch_h_plan(stride_bytes, k, nq, C.byref(w), C.byref(b)) != 0:
        return None
    return w.value, b.value


def plan_scan_shape(vtype, dim, metric):
    """(lanes per row, 16-byte chunks per lane, long-row kernel?) the plain scan would launch with - host logic only, no device"""
    lpr, u, lng = C.c_int(0), C.c_int(0), C.c_int(0)
    _check(lib().vg_plan_scan_shape(vtype, dim, metric, C.byref(lpr), C.byref(u), C.byref(lng)))
    return lpr.value, u.value, bool(lng.value)


class Corpus:
    """One HBM-resident corpus shard (opaque vg_corpus handle)."""

    def __init__(self, vtype, dim, device=0, capacity=0):
        self.h = C.c_void_p()
        self.vtype, self.dim = vtype, dim
        _check(lib().vg_corpus_create(device, vtype, dim, capacity, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().vg_corpus_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    @property
    def rows(self):
        return lib().vg_corpus_rows(self.h)

    def append(self, rows, rowids=None):
        rows = np.ascontiguousarray(rows)
        assert rows.ndim == 2 and rows.shape[1] * rows.itemsize >= self.dim * TYPE_SIZE[self.vtype]
        ids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
        _check(lib().vg_corpus_append(self.h, _ptr(rows), rows.shape[0], rows.strides[0], _ptr(ids)))

    def append_strided(self, buf, n_rows, stride, rowids=None):
        buf = np.ascontiguousarray(buf)
        ids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
        _check(lib().vg_corpus_append(self.h, _ptr(buf), n_rows, stride, _ptr(ids)))

    def append_records(self, records, n):
        records = np.ascontiguousarray(records)
        _check(lib().vg_corpus_append_records(self.h, _ptr(records), n))

    def append_device(self, dev_ptr, n_rows, stride, rowids=None):
        ids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
        _check(lib().vg_corpus_append_device(self.h, C.c_void_p(dev_ptr), n_rows, stride, _ptr(ids)))

    def set_rowid_base(self, base):
        _check(lib().vg_corpus_set_rowid_base(self.h, base))

    def scan_topk(self, metric, query, k):
        query = np.ascontiguousarray(query)
        ids = np.zeros(max(k, 1), dtype=np.int64)
        dist = np.zeros(max(k, 1), dtype=np.float64)
        cnt = C.c_int(0)
        _check(lib().vg_scan_topk(self.h, metric, _ptr(query), k, _ptr(ids), _ptr(dist), C.byref(cnt)))
        return ids[:cnt.value], dist[:cnt.value]

    def scan_topk_device(self, metric, dev_query_ptr, k, dev_keys_ptr, stream=None):
        _check(lib().vg_scan_topk_device(self.h, metric, C.c_void_p(dev_query_ptr), k, C.c_void_p(dev_keys_ptr),
                                         C.c_void_p(stream) if stream else None))

    def scan_distances(self, metric, query):
        query = np.ascontiguousarray(query)
        out = np.empty(self.rows, dtype=np.float32)
        _check(lib().vg_scan_distances(self.h, metric, _ptr(query), _ptr(out)))
        return out

    def scan_topk_batch(self, metric, queries, k):
        queries = np.ascontiguousarray(queries)
        nq = queries.shape[0]
        ids = np.zeros((nq, max(k, 1)), dtype=np.int64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        _check(lib().vg_scan_topk_batch(self.h, metric, _ptr(queries), nq, k, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def scan_topk_batch_keys(self, metric, queries, k):
        """per-query packed keys (positions local to this corpus): uint64 [nq, k] (VG_KEY_EMPTY padded), counts [nq]"""
        queries = np.ascontiguousarray(queries)
        nq = queries.shape[0]
        keys = np.full((nq, max(k, 1)), KEY_EMPTY, dtype=np.uint64)
        cnt = np.zeros(nq, dtype=np.int32)
        _check(lib().vg_scan_topk_batch_keys(self.h, metric, _ptr(queries), nq, k, _ptr(keys), _ptr(cnt)))
        for i in range(nq):                                    # entries past the count are unspecified: pad them
            keys[i, cnt[i]:] = KEY_EMPTY
        return keys, cnt

    def minmax(self):
        lo, hi, neg = C.c_float(0), C.c_float(0), C.c_int(0)
        _check(lib().vg_corpus_minmax(self.h, C.byref(lo), C.byref(hi), C.byref(neg)))
        return lo.value, hi.value, bool(neg.value)

    def quantize_rows(self, scale, offset, qtype, row0=0, n_rows=None):
        n = self.rows - row0 if n_rows is None else n_rows
        out = np.empty((n, self.dim), dtype=np.uint8)
        _check(lib().vg_corpus_quantize_rows(self.h, scale, offset, qtype, row0, n, _ptr(out)))
        return out

    def set_profiling(self, on=True):
        _check(lib().vg_set_profiling(self.h, 1 if on else 0))

    def last_kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().vg_last_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile_mean_ms(self):
        n, a, b = C.c_int(0), C.c_float(0), C.c_float(0)
        _check(lib().vg_profile_mean_ms(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value

    def kernel_name(self, metric):
        return lib().vg_scan_kernel_name(self.h, metric).decode()

    def profile_mean_ms_ex(self):
        """(launches, scan kernel ms, merge ms, pre-pass ms) - the scan figure is ONE kernel"""
        n, a, b, p = C.c_int(0), C.c_float(0), C.c_float(0), C.c_float(0)
        _check(lib().vg_profile_mean_ms_ex(self.h, C.byref(n), C.byref(a), C.byref(b), C.byref(p)))
        return n.value, a.value, b.value, p.value

    def filter_exact_evals(self):
        v = C.c_ulonglong(0)
        _check(lib().vg_filter_exact_evals(self.h, C.byref(v)))
        return v.value

    def filter_guard_cooldown(self):
        """> 0: the selectivity guard has sent this corpus' next that many single scans to the plain kernel"""
        return int(lib().vg_filter_guard_cooldown(self.h))

    def find_rowid(self, rowid):
        return int(lib().vg_corpus_find_rowid(self.h, rowid))

    def patch_rows(self, positions, rows):
        positions = np.ascontiguousarray(positions, dtype=np.int64)
        rows = np.ascontiguousarray(rows)
        _check(lib().vg_corpus_patch_rows(self.h, _ptr(positions), positions.shape[0], _ptr(rows), rows.strides[0]))

    def delete_rows(self, positions):
        positions = np.ascontiguousarray(positions, dtype=np.int64)
        _check(lib().vg_corpus_delete_rows(self.h, _ptr(positions), positions.shape[0]))

    def device_bytes(self):
        """(row matrix, derived per-row copies / statistics, working buffers) in bytes on the device"""
        out = np.zeros(3, dtype=np.int64)
        _check(lib().vg_corpus_device_bytes(self.h, _ptr(out)))
        return tuple(int(x) for x in out)

    def batch_q8_status(self):
        """how the last attempt at the int8 batch filter ended (vectorgpu_diag.h)"""
        return lib().vg_batch_q8_status(self.h)

    def last_batch_path(self):
        """1 f32 matrix-core kernel, 2 int8, 3 half-precision kernel, 4 its long-row form, 5 multi-query scan, 6 one scan per query"""
        return lib().vg_batch_last_path(self.h)

    def batch_filter_exact_evals(self):
        v = C.c_ulonglong(0)
        _check(lib().vg_batch_filter_exact_evals(self.h, C.byref(v)))
        return v.value

    def set_scan_filter(self, mode):
        _check(lib().vg_corpus_set_scan_filter(self.h, mode))

    def set_tie_order(self, mode):
        """TIE_POSITION (default) or TIE_REFERENCE: the reference's slot-history result among equal distances"""
        _check(lib().vg_corpus_set_tie_order(self.h, mode))

    def pass_ms(self, which):
        """(kernel ms, rows) of the last "minmax" / "quantize" / "q8_shadow" pass over this corpus"""
        ms, rows = C.c_float(0), C.c_longlong(0)
        _check(lib().vg_corpus_pass_ms(self.h, {"minmax": 0, "quantize": 1, "q8_shadow": 2}[which], C.byref(ms), C.byref(rows)))
        return ms.value, rows.value

    def clear(self):
        _check(lib().vg_corpus_clear(self.h))

    def tie_stats(self):
        """reference-order scans so far: {scans, with_a_tie_among_the_k_plus_1_best, fused_replays, store_mode_replays}"""
        out = np.zeros(4, dtype=np.uint64)
        _check(lib().vg_corpus_tie_stats(self.h, _ptr(out)))
        return dict(zip(("scans", "with_a_tie_among_the_k_plus_1_best", "fused_replays", "store_mode_replays"), (int(x) for x in out)))


class SlabScan:
    """One query over a table handed over slab by slab (include/vectorgpu.h: vg_slab_scan_*): the device holds two slabs."""

    def __init__(self, vtype, dim, metric, query, k, slab_rows, tie_order=0, device=0, rowid_base=1):
        self.h = C.c_void_p()
        self.k = k
        query = np.ascontiguousarray(query)
        _check(lib().vg_slab_scan_begin(device, vtype, dim, metric, _ptr(query), k, tie_order, slab_rows, rowid_base, C.byref(self.h)))

    def rows(self, rows, rowids=None):
        rows = np.ascontiguousarray(rows)
        ids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
        _check(lib().vg_slab_scan_rows(self.h, _ptr(rows), rows.shape[0], rows.strides[0] if rows.ndim == 2 else 0, None if ids is None else _ptr(ids)))

    def records(self, recs, n):
        recs = np.ascontiguousarray(recs)
        _check(lib().vg_slab_scan_records(self.h, _ptr(recs), n))

    def finish(self):
        ids = np.zeros(max(self.k, 1), dtype=np.int64)
        dist = np.zeros(max(self.k, 1), dtype=np.float64)
        cnt = C.c_int(0)
        _check(lib().vg_slab_scan_finish(self.h, _ptr(ids), _ptr(dist), C.byref(cnt)))
        return ids[:cnt.value], dist[:cnt.value]

    def all(self):
        """k = 0 scans, after finish(): (distances, rowids) of every row in scan order"""
        n = C.c_int64(0)
        pd, pi = C.c_void_p(), C.c_void_p()
        _check(lib().vg_slab_scan_all(self.h, C.byref(n), C.byref(pd), C.byref(pi)))
        if n.value == 0:
            return np.zeros(0, dtype=np.float32), np.zeros(0, dtype=np.int64)
        d = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_float)), shape=(n.value,)).copy()
        i = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_int64)), shape=(n.value,)).copy()
        return d, i

    def close(self):
        if self.h:
            lib().vg_slab_scan_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close


def device_memory(device=0):
    f, t = C.c_longlong(0), C.c_longlong(0)
    _check(lib().vg_device_memory(device, C.byref(f), C.byref(t)))
    return f.value, t.value


class Shards:
    """One logical corpus dealt block-cyclically over several devices of this process (opaque vg_shards handle)."""

    def __init__(self, vtype, dim, devices, block_rows=0):
        self.h = C.c_void_p()
        self.vtype, self.dim = vtype, dim
        devs = (C.c_int * len(devices))(*devices)
        _check(lib().vg_shards_create(devs, len(devices), vtype, dim, block_rows, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().vg_shards_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    @property
    def rows(self):
        return lib().vg_shards_rows(self.h)

    def shard_rows(self):
        return [lib().vg_corpus_rows(lib().vg_shards_shard(self.h, i)) for i in range(lib().vg_shards_count(self.h))]

    def reserve(self, n):
        _check(lib().vg_shards_reserve(self.h, n))

    def clear(self):
        _check(lib().vg_shards_clear(self.h))

    def append(self, rows, rowids=None):
        rows = np.ascontiguousarray(rows)
        ids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
        _check(lib().vg_shards_append(self.h, _ptr(rows), rows.shape[0], rows.strides[0], _ptr(ids)))

    def append_records(self, records, n):
        records = np.ascontiguousarray(records)
        _check(lib().vg_shards_append_records(self.h, _ptr(records), n))

    def rowid_at(self, pos):
        return lib().vg_shards_rowid_at(self.h, pos)

    def rowids(self, pos0, n):
        out = np.zeros(n, dtype=np.int64)
        _check(lib().vg_shards_rowids(self.h, pos0, n, _ptr(out)))
        return out

    def set_tie_order(self, mode):
        _check(lib().vg_shards_set_tie_order(self.h, mode))

    def set_scan_filter(self, mode):
        _check(lib().vg_shards_set_scan_filter(self.h, mode))

    def set_gather(self, mode):
        """candidate gather: "host" (every shard copies its 64 keys back) or "rccl" (one grouped ncclAllGather over xGMI)"""
        _check(lib().vg_shards_set_gather(self.h, {"host": 0, "rccl": 1}[mode]))

    def gather_stats(self):
        out = np.zeros(2, dtype=np.uint64)
        serving = lib().vg_shards_gather_stats(self.h, _ptr(out))
        return {"host": int(out[0]), "rccl": int(out[1]), "rccl_serving": bool(serving)}

    @property
    def threaded(self):
        """every query's per-shard work runs on the handle's persistent host threads (one per shard)"""
        return bool(lib().vg_shards_threaded(self.h))

    def tie_stats(self):
        """reference-order scans of this handle so far (same four counters as Corpus.tie_stats)"""
        out = np.zeros(4, dtype=np.uint64)
        _check(lib().vg_shards_tie_stats(self.h, _ptr(out)))
        return dict(zip(("scans", "with_a_tie_among_the_k_plus_1_best", "fused_replays", "store_mode_replays"), (int(x) for x in out)))

    def scan_topk(self, metric, query, k):
        query = np.ascontiguousarray(query)
        ids = np.zeros(max(k, 1), dtype=np.int64)
        dist = np.zeros(max(k, 1), dtype=np.float64)
        cnt = C.c_int(0)
        _check(lib().vg_shards_scan_topk(self.h, metric, _ptr(query), k, _ptr(ids), _ptr(dist), C.byref(cnt)))
        return ids[:cnt.value], dist[:cnt.value]

    def scan_topk_batch(self, metric, queries, k):
        queries = np.ascontiguousarray(queries)
        nq = queries.shape[0]
        ids = np.zeros((nq, max(k, 1)), dtype=np.int64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        _check(lib().vg_shards_scan_topk_batch(self.h, metric, _ptr(queries), nq, k, _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def scan_distances(self, metric, query):
        query = np.ascontiguousarray(query)
        out = np.empty(self.rows, dtype=np.float32)
        _check(lib().vg_shards_scan_distances(self.h, metric, _ptr(query), _ptr(out)))
        return out

    def minmax(self):
        lo, hi, neg = C.c_float(0), C.c_float(0), C.c_int(0)
        _check(lib().vg_shards_minmax(self.h, C.byref(lo), C.byref(hi), C.byref(neg)))
        return lo.value, hi.value, bool(neg.value)

    def quantize_rows(self, scale, offset, qtype, row0=0, n_rows=None):
        n = self.rows - row0 if n_rows is None else n_rows
        out = np.empty((n, self.dim), dtype=np.uint8)
        _check(lib().vg_shards_quantize_rows(self.h, scale, offset, qtype, row0, n, _ptr(out)))
        return out


def quantize_query(src_type, src, scale, offset, qtype):
    src = np.ascontiguousarray(src)
    dst = np.empty(src.shape[0], dtype=np.uint8 if qtype == QUANT_U8 else np.int8)
    _check(lib().vg_quantize_query(src_type, _ptr(src), src.shape[0], scale, offset, qtype, _ptr(dst)))
    return dst


def reference_topk_replay(dist, k, below_cap=0):
    """host replay of the reference's slot algorithm over a distance stream -> (positions, distances)"""
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    pos = np.zeros(max(k, 1), dtype=np.int64)
    out = np.zeros(max(k, 1), dtype=np.float64)
    cnt = lib().vg_reference_topk_replay(_ptr(dist), dist.shape[0], k, below_cap, _ptr(pos), _ptr(out))
    if cnt < 0:
        raise VectorGpuError("vg_reference_topk_replay: bad arguments")
    return pos[:cnt], out[:cnt]


def reference_topk_replay_slabs(dist, k, slab_rows, below_cap=0):
    """the same stream handed over slab by slab (the out-of-core scan's continuation of the slot state) -> (positions, distances)"""
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    pos = np.zeros(max(k, 1), dtype=np.int64)
    out = np.zeros(max(k, 1), dtype=np.float64)
    cnt = lib().vg_reference_topk_replay_slabs(_ptr(dist), dist.shape[0], k, slab_rows, below_cap, _ptr(pos), _ptr(out))
    if cnt < 0:
        raise VectorGpuError("vg_reference_topk_replay_slabs: bad arguments")
    return pos[:cnt], out[:cnt]


def merge_keys_batch(keys, pos_offsets, k):
    """keys: (n_lists, nq, list_len) uint64 -> (global positions [nq, k], distances [nq, k], counts [nq])."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n_lists, nq, list_len = keys.shape
    off = None if pos_offsets is None else np.ascontiguousarray(pos_offsets, dtype=np.int64)
    pos = np.zeros((nq, max(k, 1)), dtype=np.int64)
    dist = np.zeros((nq, max(k, 1)), dtype=np.float64)
    cnt = np.zeros(nq, dtype=np.int32)
    if lib().vg_merge_keys_batch(_ptr(keys), n_lists, nq, list_len, _ptr(off), k, _ptr(pos), _ptr(dist), _ptr(cnt)) != 0:
        raise VectorGpuError("vg_merge_keys_batch: bad arguments")
    return pos, dist, cnt


def merge_keys(keys, pos_offsets, k):
    """keys: (n_lists, list_len) uint64 array of per-shard candidate lists -> (global positions, distances)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n_lists, list_len = keys.shape
    off = None if pos_offsets is None else np.ascontiguousarray(pos_offsets, dtype=np.int64)
    pos = np.zeros(max(k, 1), dtype=np.int64)
    dist = np.zeros(max(k, 1), dtype=np.float64)
    cnt = lib().vg_merge_keys(_ptr(keys), n_lists, list_len, _ptr(off), k, _ptr(pos), _ptr(dist))
    return pos[:cnt], dist[:cnt]
