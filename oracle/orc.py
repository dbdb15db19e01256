"""ctypes front-end for the test-only CPU oracle (oracle/liboracle.so) and, when built,
the reference's own code (oracle/_ref/*.so, compiled from /root/reference by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (sqlite-vector_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

# enums, mirror /root/reference/src/distance-cpu.h:36-58
F32, F16, BF16, U8, I8 = 1, 2, 3, 4, 5
L2, SQUARED_L2, COSINE, DOT, L1 = 1, 2, 3, 4, 5
QUANT_U8, QUANT_S8 = 1, 2
CPU, AVX2 = 0, 1

TYPE_NAMES = {F32: "f32", F16: "f16", BF16: "bf16", U8: "u8", I8: "i8"}
METRIC_NAMES = {L2: "l2", SQUARED_L2: "sql2", COSINE: "cosine", DOT: "dot", L1: "l1"}
TYPE_SIZE = {F32: 4, F16: 2, BF16: 2, U8: 1, I8: 1}
NP_DTYPE = {F32: np.float32, F16: np.uint16, BF16: np.uint16, U8: np.uint8, I8: np.int8}


def build(ref=True):
    """Compile liboracle.so (always) and oracle/_ref (only when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", HERE, "all" if ref else os.path.join(HERE, "liboracle.so")],
                   check=True)


def _lib():
    path = os.path.join(HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.orc_distance.restype = C.c_float
    lib.orc_distance.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_clamp.restype = C.c_float
    lib.orc_clamp.argtypes = [C.c_float]
    lib.orc_scan_distances.restype = None
    lib.orc_scan_distances.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int64, C.c_int, C.c_void_p]
    lib.orc_topk_reference.restype = C.c_int
    lib.orc_topk_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_topk_ordered.restype = C.c_int
    lib.orc_topk_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    lib.orc_scan_topk_reference.restype = C.c_int
    lib.orc_scan_topk_reference.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_scan_topk_with_fn.restype = C.c_int
    lib.orc_scan_topk_with_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_quantize.restype = None
    lib.orc_scan_topk_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_scan_topk_threads.restype = C.c_int
    lib.orc_scan_topk_threads_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                                 C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
    lib.orc_scan_topk_threads_pinned.restype = C.c_int
    lib.orc_quantize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int]
    lib.orc_quant_params.restype = None
    lib.orc_quant_params.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.orc_f16_to_f32.restype = C.c_float
    lib.orc_f16_to_f32.argtypes = [C.c_uint16]
    lib.orc_f32_to_f16.restype = C.c_uint16
    lib.orc_f32_to_f16.argtypes = [C.c_float]
    lib.orc_bf16_to_f32.restype = C.c_float
    lib.orc_bf16_to_f32.argtypes = [C.c_uint16]
    lib.orc_f32_to_bf16.restype = C.c_uint16
    lib.orc_f32_to_bf16.argtypes = [C.c_float]
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _lib()
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def distance(backend, metric, vtype, a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return float(lib().orc_distance(backend, metric, vtype, _ptr(a), _ptr(b), a.shape[0]))


def scan_distances(backend, metric, vtype, query, rows):
    """rows: C-contiguous (n, dim) array of the storage dtype. Returns float32 (n,) incl. the clamp."""
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    n, dim = rows.shape
    out = np.empty(n, dtype=np.float32)
    lib().orc_scan_distances(backend, metric, vtype, _ptr(query), _ptr(rows), n, rows.strides[0], dim, _ptr(out))
    return out


def topk_reference(dist, rowids, k):
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    rowids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
    out_ids = np.zeros(max(k, 1), dtype=np.int64)
    out_d = np.zeros(max(k, 1), dtype=np.float64)
    cnt = lib().orc_topk_reference(_ptr(dist), _ptr(rowids), dist.shape[0], k, _ptr(out_ids), _ptr(out_d))
    return out_ids[:cnt], out_d[:cnt]


def topk_ordered(dist, rowids, k):
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    rowids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
    out_ids = np.zeros(max(k, 1), dtype=np.int64)
    out_d = np.zeros(max(k, 1), dtype=np.float64)
    out_p = np.zeros(max(k, 1), dtype=np.int64)
    cnt = lib().orc_topk_ordered(_ptr(dist), _ptr(rowids), dist.shape[0], k, _ptr(out_ids), _ptr(out_d), _ptr(out_p))
    return out_ids[:cnt], out_d[:cnt], out_p[:cnt]


def scan_topk_reference(backend, metric, vtype, query, rows, rowids, k):
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    rowids = None if rowids is None else np.ascontiguousarray(rowids, dtype=np.int64)
    n, dim = rows.shape
    out_ids = np.zeros(max(k, 1), dtype=np.int64)
    out_d = np.zeros(max(k, 1), dtype=np.float64)
    cnt = lib().orc_scan_topk_reference(backend, metric, vtype, _ptr(query), _ptr(rows), n, rows.strides[0], dim,
                                        _ptr(rowids), k, _ptr(out_ids), _ptr(out_d))
    return out_ids[:cnt], out_d[:cnt]


def quantize(vtype, src, offset, scale, qtype):
    src = np.ascontiguousarray(src)
    n = src.shape[-1] if src.ndim == 1 else src.size
    dst = np.empty(n, dtype=np.uint8 if qtype == QUANT_U8 else np.int8)
    lib().orc_quantize(vtype, _ptr(src), _ptr(dst), C.c_float(offset), C.c_float(scale), n, qtype)
    return dst


def quant_params(vtype, rows, qtype=0):
    rows = np.ascontiguousarray(rows)
    n, dim = rows.shape
    qt = C.c_int(qtype)
    sc = C.c_float(0)
    off = C.c_float(0)
    lib().orc_quant_params(vtype, _ptr(rows), n, rows.strides[0], dim, C.byref(qt), C.byref(sc), C.byref(off))
    return qt.value, sc.value, off.value


# ---------------------------------------------------------------- numpy helpers for 16-bit types

def f32_to_bf16_bits(x):
    """RNE, same formula as distance-cpu.h:103-108 (vectorised)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    rnd = 0x7FFF + ((u >> 16) & 1)
    return ((u + rnd) >> 16).astype(np.uint16)


def f32_to_f16_bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)


def to_storage(vtype, x_f32):
    """float32 array -> the storage representation of vtype (bits for f16/bf16)."""
    if vtype == F32:
        return np.ascontiguousarray(x_f32, dtype=np.float32)
    if vtype == F16:
        return f32_to_f16_bits(x_f32)
    if vtype == BF16:
        return f32_to_bf16_bits(x_f32)
    if vtype == U8:
        return np.clip(np.rint(x_f32), 0, 255).astype(np.uint8)
    if vtype == I8:
        return np.clip(np.rint(x_f32), -128, 127).astype(np.int8)
    raise ValueError(vtype)


# ---------------------------------------------------------------- the reference itself (oracle/_ref)

class RefKernels:
    """dispatch_distance_table of the reference (distance-cpu.c:21), loaded from oracle/_ref/libref_{cpu,avx2}.so."""

    def __init__(self, which):
        path = os.path.join(REF_DIR, "libref_%s.so" % which)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.init_distance_functions.argtypes = [C.c_bool]
        self.lib.init_distance_functions.restype = None
        self.lib.init_distance_functions(False)          # exactly what sqlite3_vector_init does (:2561)
        fn_t = C.CFUNCTYPE(C.c_float, C.c_void_p, C.c_void_p, C.c_int)
        table_t = (fn_t * 6) * 6
        self.table = table_t.in_dll(self.lib, "dispatch_distance_table")
        self.backend_name = C.c_char_p.in_dll(self.lib, "distance_backend_name").value.decode()

    def distance(self, metric, vtype, a, b):
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        return float(self.table[metric][vtype](_ptr(a), _ptr(b), a.shape[0]))

    def scan_topk(self, metric, vtype, query, rows, k):
        """the reference's kernel (function pointer out of its dispatch table) inside the reference's top-k loop"""
        rows = np.ascontiguousarray(rows)
        query = np.ascontiguousarray(query)
        fnptr = C.cast(self.table[metric][vtype], C.c_void_p)
        out_ids = np.zeros(max(k, 1), dtype=np.int64)
        out_d = np.zeros(max(k, 1), dtype=np.float64)
        cnt = lib().orc_scan_topk_with_fn(fnptr, _ptr(query), _ptr(rows), rows.shape[0], rows.strides[0],
                                          rows.shape[1], None, k, _ptr(out_ids), _ptr(out_d))
        return out_ids[:cnt], out_d[:cnt]

    def scan_topk_all_cores(self, metric, vtype, queries, rows, k, nthreads, seconds, cpus=None, repeat=1):
        """the reference's kernel inside the reference's top-k loop on `nthreads` pthreads, a row split of `rows` (oracle.c:
        orc_scan_topk_threads_pinned): thread i pinned to logical CPU cpus[i] (None: unpinned), its private copy holding its range `repeat`
        times over; returns (vectors scanned per second, rows per thread, per-range lists of query 0: ids, dist, counts, threads pinned)"""
        rows = np.ascontiguousarray(rows)
        queries = np.ascontiguousarray(queries)
        fnptr = C.cast(self.table[metric][vtype], C.c_void_p)
        first_ids = np.zeros((nthreads, k), dtype=np.int64)
        first_d = np.zeros((nthreads, k), dtype=np.float64)
        counts = np.zeros(nthreads, dtype=np.int32)
        per, el, scans, pinned = C.c_int64(0), C.c_double(0.0), C.c_int64(0), C.c_int(0)
        cpu_arr = np.ascontiguousarray(np.asarray(cpus, dtype=np.int32)) if cpus is not None else None
        if cpu_arr is not None and cpu_arr.shape[0] < nthreads:
            raise ValueError("cpus: one logical CPU per thread")
        rc = lib().orc_scan_topk_threads_pinned(fnptr, _ptr(queries), queries.shape[0], queries.strides[0], _ptr(rows), rows.shape[0], rows.strides[0],
                                                rows.shape[1], k, nthreads, float(seconds), _ptr(cpu_arr) if cpu_arr is not None else None, int(repeat),
                                                _ptr(first_ids), _ptr(first_d), _ptr(counts), C.byref(per), C.byref(el), C.byref(scans),
                                                C.byref(pinned))
        if rc != 0:
            raise RuntimeError("orc_scan_topk_threads_pinned: %d" % rc)
        return scans.value * per.value / el.value, per.value, first_ids, first_d, counts, pinned.value

    def scan(self, metric, vtype, query, rows):
        rows = np.ascontiguousarray(rows)
        query = np.ascontiguousarray(query)
        fn = self.table[metric][vtype]
        qp = _ptr(query)
        base = rows.ctypes.data
        stride = rows.strides[0]
        dim = rows.shape[1]
        out = np.empty(rows.shape[0], dtype=np.float32)
        for r in range(rows.shape[0]):
            out[r] = fn(qp, C.c_void_p(base + r * stride), dim)
        return out


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libref_cpu.so")) and \
        os.path.exists(os.path.join(REF_DIR, "libref_avx2.so"))


def ref_extension_path(which):
    """Path (without .so, as sqlite's load_extension wants) of the reference extension build."""
    p = os.path.join(REF_DIR, which, "vector.so")
    return p[:-3] if os.path.exists(p) else None
