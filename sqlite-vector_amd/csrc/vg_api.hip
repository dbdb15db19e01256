// vg_api.hip - implementation of the C-ABI in include/vectorgpu.h on top of the gfx950 kernels.
//
// Host-side responsibilities only: device memory for the staged corpus, query upload, kernel selection and
// launch, decoding the k winning keys into (rowid, distance).  No distance is ever computed on the host: if
// the HIP runtime / a gfx950 device is missing every entry point fails with VG_ERR_NO_DEVICE.
#include "../../include/vectorgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vg_scan.h"

#define VG_PROF_RING 1024

// ------------------------------------------------------------------------------------------------ errors

static thread_local std::string g_err;

static int vg_fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return vg_fail(e__ == hipErrorOutOfMemory ? VG_ERR_NOMEM : VG_ERR_HIP, "%s failed: %s (%s:%d)", \
                           #expr, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)

extern "C" const char *vg_last_error(void) { return g_err.c_str(); }
extern "C" void vg_set_last_error_(const char *msg) { g_err = msg ? msg : ""; }     // for vg_shards.hip

extern "C" int vg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *vg_backend_name(void) {
    static char name[128] = {0};
    if (name[0]) return name;
    int n = vg_device_count();
    if (n <= 0) {
        snprintf(name, sizeof(name), "HIP (no device)");
        return name;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) == hipSuccess) {
        char arch[64];
        snprintf(arch, sizeof(arch), "%s", p.gcnArchName);
        char *colon = strchr(arch, ':');
        if (colon) *colon = 0;
        snprintf(name, sizeof(name), "HIP %s x%d", arch, n);
    } else {
        snprintf(name, sizeof(name), "HIP");
    }
    return name;
}

// ------------------------------------------------------------------------------------------------ corpus

static int elem_size(int vtype) {
    switch (vtype) {
        case VG_TYPE_F32: return 4;
        case VG_TYPE_F16: case VG_TYPE_BF16: return 2;
        case VG_TYPE_U8: case VG_TYPE_I8: return 1;
    }
    return 0;
}

struct vg_corpus {
    int device = 0;
    int vtype = 0;
    int dim = 0;
    int es = 0;
    int nch = 0;               // 16-byte chunks per stored row
    int64_t stride = 0;        // bytes per stored row (nch * 16)
    int64_t n_rows = 0;
    int64_t cap_rows = 0;
    uint8_t *d_rows = nullptr;
    std::vector<int64_t> rowids;   // empty => implicit rowid_base + position
    int64_t rowid_base = 1;

    hipStream_t stream = nullptr;
    uint8_t *d_query = nullptr;    // nch*16 bytes
    uint8_t *h_query = nullptr;    // pinned
    uint64_t *d_cand = nullptr;    // max_blocks * 64 keys
    uint64_t *d_keys = nullptr;    // 64 keys
    uint64_t *h_keys = nullptr;    // pinned, 64 keys
    float *d_dist = nullptr;       // lazily sized to n_rows (stream scans / large k)
    int64_t d_dist_cap = 0;
    uint64_t *d_sel_keys = nullptr, *d_sel_sorted = nullptr;   // k > 64 path: N keys, unsorted / sorted
    void *d_sel_temp = nullptr;
    uint32_t *d_sel_state = nullptr;   // radix-select state + histogram (vg_select.hip)
    size_t sel_temp_bytes = 0;
    int64_t sel_cap = 0;
    uint8_t *pin[2] = {nullptr, nullptr};      // staging pipeline: pinned bounce buffers + their completion events
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    int pin_idx = 0;
    uint8_t *d_stage = nullptr;                // device-side landing zone for rows that need de-interleaving
    hipEvent_t append_ev = nullptr;            // recorded behind the last enqueued host append (other streams wait on it)
    bool append_pending = false;
    bool enqueued = false;                     // a vg_scan_topk_enqueue is in flight (vg_scan_topk_collect pending)
    float *d_xnorm = nullptr;                  // lazily: row norms for rows [0, xnorm_rows) (see ensure_row_norms)
    int64_t xnorm_rows = 0, xnorm_cap = 0;
    hipEvent_t norm_ev = nullptr;
    // quantized batches (vg_batch_i8.hip): per-row sum x / sum x^2 and, for uint8, the XOR-0x80 copy the matrix core reads
    int32_t *d_sx = nullptr;
    uint32_t *d_sxx = nullptr;
    uint8_t *d_rows_s8 = nullptr;
    int64_t i8_rows = 0, i8_cap = 0;              // orders a caller-stream scan behind a norm pass on the corpus stream
    void *d_bq = nullptr;          // batched path: padded queries, per-(query, partition) candidates, final keys
    uint64_t *d_bcand = nullptr, *d_bkeys = nullptr;
    size_t bq_bytes = 0, bcand_bytes = 0, bkeys_bytes = 0;
    int max_blocks = 0;
    int cu_count = 0;

    // instrumentation: a ring of event triples (before scan | after scan | after merge), recorded on the stream
    // each launch runs on, read back only when asked - no host synchronisation inside a timed region
    bool profiling = false;
    std::vector<hipEvent_t> ev;            // 3 * VG_PROF_RING events, created on first enable
    std::vector<uint8_t> ev_had_merge;
    long long prof_launches = 0;           // launches recorded since profiling was (re)enabled
    float last_scan_ms = 0.f, last_merge_ms = 0.f;
    char kernel_name[64] = {0};
};

static int env_int(const char *name, int dflt) {
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    return atoi(s);
}

extern "C" int vg_corpus_create(int device, int vtype, int dim, int64_t capacity_rows_hint, vg_corpus **out) {
    if (!out) return vg_fail(VG_ERR_INVALID, "vg_corpus_create: out is NULL");
    *out = nullptr;
    int es = elem_size(vtype);
    if (es == 0) return vg_fail(VG_ERR_INVALID, "vg_corpus_create: unknown vector type %d", vtype);
    if (dim <= 0) return vg_fail(VG_ERR_INVALID, "vg_corpus_create: dimension must be positive (got %d)", dim);
    int ndev = vg_device_count();
    if (ndev <= 0) return vg_fail(VG_ERR_NO_DEVICE, "no HIP device available (the scan path is GPU-only)");
    if (device < 0 || device >= ndev) return vg_fail(VG_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    int64_t row_bytes = (int64_t)dim * es;
    if (row_bytes > 128 * 1024) return vg_fail(VG_ERR_UNSUPPORTED, "rows larger than 128 KiB are not supported (dim=%d): the query must fit the CU's 160 KiB LDS", dim);
    HIP_TRY(hipSetDevice(device));
    vg_corpus *c = new vg_corpus();
    c->device = device;
    c->vtype = vtype;
    c->dim = dim;
    c->es = es;
    c->nch = (int)((row_bytes + 15) / 16);
    c->stride = (int64_t)c->nch * 16;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) { delete c; return vg_fail(VG_ERR_HIP, "hipGetDeviceProperties failed"); }
    c->cu_count = p.multiProcessorCount;
    c->max_blocks = c->cu_count * 8;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc(&c->d_query, (size_t)c->stride)) != hipSuccess ||
        (e = hipHostMalloc(&c->h_query, (size_t)c->stride)) != hipSuccess ||
        (e = hipMalloc(&c->d_cand, (size_t)c->max_blocks * VG_WAVE * sizeof(uint64_t))) != hipSuccess ||
        (e = hipMalloc(&c->d_keys, VG_WAVE * sizeof(uint64_t))) != hipSuccess ||
        (e = hipHostMalloc(&c->h_keys, VG_WAVE * sizeof(uint64_t))) != hipSuccess) {
        vg_corpus_destroy(c);
        return vg_fail(VG_ERR_HIP, "vg_corpus_create: device setup failed: %s", hipGetErrorString(e));
    }
    if (capacity_rows_hint > 0) {
        e = hipMalloc(&c->d_rows, (size_t)(capacity_rows_hint * c->stride));
        if (e != hipSuccess) {
            vg_corpus_destroy(c);
            return vg_fail(VG_ERR_NOMEM, "vg_corpus_create: cannot allocate %lld bytes of HBM: %s",
                           (long long)(capacity_rows_hint * c->stride), hipGetErrorString(e));
        }
        c->cap_rows = capacity_rows_hint;
    }
    *out = c;
    return VG_OK;
}

extern "C" void vg_corpus_destroy(vg_corpus *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->d_rows) hipFree(c->d_rows);
    if (c->d_query) hipFree(c->d_query);
    if (c->h_query) hipHostFree(c->h_query);
    if (c->d_cand) hipFree(c->d_cand);
    if (c->d_keys) hipFree(c->d_keys);
    if (c->h_keys) hipHostFree(c->h_keys);
    if (c->d_dist) hipFree(c->d_dist);
    if (c->d_sel_keys) hipFree(c->d_sel_keys);
    if (c->d_sel_sorted) hipFree(c->d_sel_sorted);
    if (c->d_sel_temp) hipFree(c->d_sel_temp);
    for (int i = 0; i < 2; ++i) { if (c->pin[i]) hipHostFree(c->pin[i]); if (c->pin_ev[i]) hipEventDestroy(c->pin_ev[i]); }
    if (c->append_ev) hipEventDestroy(c->append_ev);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->d_bq) hipFree(c->d_bq);
    if (c->d_xnorm) hipFree(c->d_xnorm);
    if (c->d_sel_state) hipFree(c->d_sel_state);
    if (c->d_sx) hipFree(c->d_sx);
    if (c->d_sxx) hipFree(c->d_sxx);
    if (c->d_rows_s8) hipFree(c->d_rows_s8);
    if (c->norm_ev) hipEventDestroy(c->norm_ev);
    if (c->d_bcand) hipFree(c->d_bcand);
    if (c->d_bkeys) hipFree(c->d_bkeys);
    for (hipEvent_t e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int vg_corpus_clear(vg_corpus *c) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    c->n_rows = 0;
    c->xnorm_rows = 0;
    c->i8_rows = 0;
    c->rowids.clear();
    return VG_OK;
}

extern "C" int64_t vg_corpus_rows(const vg_corpus *c) { return c ? c->n_rows : 0; }
extern "C" int vg_corpus_dim(const vg_corpus *c) { return c ? c->dim : 0; }
extern "C" int vg_corpus_type(const vg_corpus *c) { return c ? c->vtype : 0; }
extern "C" int vg_corpus_device(const vg_corpus *c) { return c ? c->device : -1; }
extern "C" int64_t vg_corpus_hbm_bytes(const vg_corpus *c) { return c ? c->cap_rows * c->stride : 0; }
extern "C" int vg_corpus_set_rowid_base(vg_corpus *c, int64_t base) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    c->rowid_base = base;
    return VG_OK;
}
extern "C" int64_t vg_corpus_rowid_at(const vg_corpus *c, int64_t position) {
    if (!c || position < 0 || position >= c->n_rows) return 0;
    return c->rowids.empty() ? c->rowid_base + position : c->rowids[(size_t)position];
}

static int corpus_reserve(vg_corpus *c, int64_t need_rows) {
    if (need_rows <= c->cap_rows) return VG_OK;
    if (need_rows >= (1ll << 32)) return vg_fail(VG_ERR_UNSUPPORTED, "a corpus shard holds at most 2^32-1 rows");
    int64_t new_cap = std::max<int64_t>(need_rows, c->cap_rows + c->cap_rows / 2);
    new_cap = std::max<int64_t>(new_cap, 1024);
    uint8_t *nb = nullptr;
    HIP_TRY(hipMalloc(&nb, (size_t)(new_cap * c->stride)));
    if (c->n_rows > 0) {
        hipError_t e = hipMemcpyAsync(nb, c->d_rows, (size_t)(c->n_rows * c->stride), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { hipFree(nb); return vg_fail(VG_ERR_HIP, "corpus grow copy failed: %s", hipGetErrorString(e)); }
    }
    if (c->d_rows) hipFree(c->d_rows);
    c->d_rows = nb;
    c->cap_rows = new_cap;
    return VG_OK;
}

extern "C" int vg_corpus_reserve(vg_corpus *c, int64_t capacity_rows) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return corpus_reserve(c, capacity_rows);
}

static void note_rowids(vg_corpus *c, const int64_t *rowids, int64_t n) {
    if (rowids) {
        if (c->rowids.empty() && c->n_rows > 0) {
            c->rowids.resize((size_t)c->n_rows);
            for (int64_t i = 0; i < c->n_rows; ++i) c->rowids[(size_t)i] = c->rowid_base + i;
        }
        c->rowids.insert(c->rowids.end(), rowids, rowids + n);
    } else if (!c->rowids.empty()) {
        for (int64_t i = 0; i < n; ++i) c->rowids.push_back(c->rowid_base + c->n_rows + i);
    }
}

// De-interleave / pad: src rows (byte stride src_stride, payload at src_off, row_bytes long) -> 16-byte-multiple
// rows.  One thread per destination 16-byte chunk; byte gathers because the source is arbitrarily aligned
// (the reference's quantized records have a stride of 8+dim).
__global__ void vg_repack_kernel(const uint8_t *src, long long src_stride, int src_off, int row_bytes,
                                 uint8_t *dst, long long dst_stride, int nch, long long n_rows) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = n_rows * nch;
    if (t >= total) return;
    long long r = t / nch;
    int ch = (int)(t - r * nch);
    const uint8_t *s = src + r * src_stride + src_off + (long long)ch * 16;
    int remain = row_bytes - ch * 16;
    uint32_t w[4] = {0, 0, 0, 0};
    if (remain >= 16 && ((reinterpret_cast<uintptr_t>(s) & 3) == 0)) {
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s);
        w[0] = s4[0]; w[1] = s4[1]; w[2] = s4[2]; w[3] = s4[3];
    } else {
        int nb = remain < 16 ? remain : 16;
        for (int j = 0; j < nb; ++j) w[j >> 2] |= (uint32_t)s[j] << ((j & 3) * 8);
    }
    *reinterpret_cast<uint4 *>(dst + r * dst_stride + (long long)ch * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// Host -> HBM staging pipeline: two pinned bounce buffers.  The caller's rows are memcpy'd into a pinned buffer and
// the H2D copy (plus, when the layouts differ, the de-interleave kernel) is only ENQUEUED on the corpus stream, so
// the call returns while the transfer runs and the caller's next sqlite3_step() batch overlaps with it.  A buffer is
// reused only after the event recorded behind its last copy has fired.  Scans run on the same stream: ordered.
#define VG_PIN_BYTES (16ll << 20)

static int pin_acquire(vg_corpus *c, uint8_t **buf, int *slot) {
    if (!c->pin[0]) {
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipHostMalloc(&c->pin[i], (size_t)VG_PIN_BYTES));
            HIP_TRY(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
        }
        HIP_TRY(hipMalloc(&c->d_stage, (size_t)VG_PIN_BYTES));
        HIP_TRY(hipEventCreateWithFlags(&c->append_ev, hipEventDisableTiming));
    }
    *slot = c->pin_idx;
    c->pin_idx ^= 1;
    if (c->pin_busy[*slot]) { HIP_TRY(hipEventSynchronize(c->pin_ev[*slot])); c->pin_busy[*slot] = false; }
    *buf = c->pin[*slot];
    return VG_OK;
}

// copies [n_rows x src_stride] host or device bytes into the padded matrix at the current end of the corpus
static int append_impl(vg_corpus *c, const void *src, bool src_on_device, int64_t n_rows, int64_t src_stride,
                       int src_off) {
    const int64_t row_bytes = (int64_t)c->dim * c->es;
    HIP_TRY(hipSetDevice(c->device));
    int rc = corpus_reserve(c, c->n_rows + n_rows);
    if (rc != VG_OK) return rc;
    uint8_t *dst = c->d_rows + c->n_rows * c->stride;
    // a plain copy is only valid when source rows have no padding of their own: padding bytes must be ZERO in HBM
    // (they are summed like data), so any row whose size is not a 16-byte multiple goes through the repack kernel
    const bool same_layout = (src_off == 0 && src_stride == c->stride && row_bytes == c->stride);
    if (src_on_device) {
        if (same_layout) {
            HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(n_rows * c->stride), hipMemcpyDeviceToDevice, c->stream));
        } else {
            long long total = n_rows * c->nch;
            hipLaunchKernelGGL(vg_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                               (const uint8_t *)src, (long long)src_stride, src_off, (int)row_bytes, dst,
                               (long long)c->stride, c->nch, (long long)n_rows);
        }
        HIP_TRY(hipStreamSynchronize(c->stream));          // the caller may free / overwrite its device buffer
        return VG_OK;
    }
    if (src_stride > VG_PIN_BYTES) return vg_fail(VG_ERR_UNSUPPORTED, "row stride %lld exceeds the staging buffer", (long long)src_stride);
    const int64_t piece_rows = std::max<int64_t>(1, VG_PIN_BYTES / src_stride);
    for (int64_t r0 = 0; r0 < n_rows; r0 += piece_rows) {
        const int64_t nr = std::min(piece_rows, n_rows - r0);
        const uint8_t *s = (const uint8_t *)src + r0 * src_stride;
        // the last row may be shorter than the stride in the caller's buffer: copy only what is addressable
        const size_t bytes = (size_t)((nr - 1) * src_stride + src_off + row_bytes);
        uint8_t *pin;
        int slot;
        rc = pin_acquire(c, &pin, &slot);
        if (rc != VG_OK) return rc;
        memcpy(pin, s, bytes);
        if (same_layout) {
            HIP_TRY(hipMemcpyAsync(dst + r0 * c->stride, pin, bytes, hipMemcpyHostToDevice, c->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(c->d_stage, pin, bytes, hipMemcpyHostToDevice, c->stream));
            long long total = nr * c->nch;
            hipLaunchKernelGGL(vg_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                               (const uint8_t *)c->d_stage, (long long)src_stride, src_off, (int)row_bytes,
                               dst + r0 * c->stride, (long long)c->stride, c->nch, (long long)nr);
        }
        HIP_TRY(hipEventRecord(c->pin_ev[slot], c->stream));
        c->pin_busy[slot] = true;
    }
    HIP_TRY(hipEventRecord(c->append_ev, c->stream));
    c->append_pending = true;
    HIP_TRY(hipGetLastError());
    return VG_OK;
}

extern "C" int vg_corpus_append(vg_corpus *c, const void *host_rows, int64_t n_rows, int64_t row_stride_bytes,
                                const int64_t *rowids) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    if (n_rows == 0) return VG_OK;
    if (!host_rows || n_rows < 0) return vg_fail(VG_ERR_INVALID, "vg_corpus_append: bad rows pointer / count");
    if (row_stride_bytes < (int64_t)c->dim * c->es) return vg_fail(VG_ERR_INVALID, "vg_corpus_append: stride %lld smaller than a row (%lld bytes)", (long long)row_stride_bytes, (long long)c->dim * c->es);
    int rc = append_impl(c, host_rows, false, n_rows, row_stride_bytes, 0);
    if (rc != VG_OK) return rc;
    note_rowids(c, rowids, n_rows);
    c->n_rows += n_rows;
    return VG_OK;
}

extern "C" int vg_corpus_append_device(vg_corpus *c, const void *dev_rows, int64_t n_rows, int64_t row_stride_bytes,
                                       const int64_t *host_rowids) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    if (n_rows == 0) return VG_OK;
    if (!dev_rows || n_rows < 0) return vg_fail(VG_ERR_INVALID, "vg_corpus_append_device: bad rows pointer / count");
    if (row_stride_bytes < (int64_t)c->dim * c->es) return vg_fail(VG_ERR_INVALID, "vg_corpus_append_device: stride smaller than a row");
    int rc = append_impl(c, dev_rows, true, n_rows, row_stride_bytes, 0);
    if (rc != VG_OK) return rc;
    note_rowids(c, host_rowids, n_rows);
    c->n_rows += n_rows;
    return VG_OK;
}

extern "C" int vg_corpus_append_records(vg_corpus *c, const void *host_records, int64_t n_records) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    if (c->vtype != VG_TYPE_U8 && c->vtype != VG_TYPE_I8) return vg_fail(VG_ERR_INVALID, "vg_corpus_append_records: corpus must be UINT8 or INT8");
    if (n_records == 0) return VG_OK;
    if (!host_records || n_records < 0) return vg_fail(VG_ERR_INVALID, "vg_corpus_append_records: bad pointer / count");
    const int64_t rec = 8 + (int64_t)c->dim;
    int rc = append_impl(c, host_records, false, n_records, rec, 8);
    if (rc != VG_OK) return rc;
    // rowids: little-endian int64 in front of every record (sqlite-vector.c:86-94, INT64_FROM_INT8PTR)
    std::vector<int64_t> ids((size_t)n_records);
    const uint8_t *p = (const uint8_t *)host_records;
    for (int64_t i = 0; i < n_records; ++i) {
        const uint8_t *q = p + i * rec;
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v |= (uint64_t)q[b] << (8 * b);
        ids[(size_t)i] = (int64_t)v;
    }
    note_rowids(c, ids.data(), n_records);
    c->n_rows += n_records;
    return VG_OK;
}

// ------------------------------------------------------------------------------------------------ kernel selection

typedef void (*scan_fn_t)(ScanArgs);

struct Shape { int lpr_log2; int U; bool long_rows; };

static const int kAllowedU[] = {1, 2, 3, 4, 6, 8};

// Pick (lanes per row, chunks per lane): cover nch chunks with lpr*U slots, wasting as few lane slots as
// possible; prefer >= 128 contiguous bytes per row per load instruction, then U = 6/4/3 (bytes in flight per lane).
// Rows that no (lpr <= 64, U <= cap) shape covers take the long-row kernel (query in LDS, one row per wavefront).
// f16 / bf16: the f64 arithmetic (4 f64 accumulators + the widening temporaries) leaves room for fewer chunks per lane
// under the 128-VGPR cap of 16 wavefronts per CU.  The query stays in registers as RAW halves (vg_scan.h launders it
// every batch so the compiler cannot hoist widened copies out of the loop); with that U = 6 fits for everything but
// bf16 dot / cosine, which spill beyond U = 3 (measured 2.4 TB/s at U = 6).
static int max_chunks_per_lane(int vtype, int acc) {
    if (vtype == VG_TYPE_F16) return 6;
    if (vtype == VG_TYPE_BF16) return (acc == A_L2 || acc == A_L1) ? 6 : 3;
    return 8;
}

static bool choose_shape(int nch, int vtype, int acc, Shape *out) {
    static const int pref[9] = {0, 1, 2, 4, 5, 0, 6, 0, 3};   // preference rank by U (higher is better)
    const int max_u = max_chunks_per_lane(vtype, acc);
    double best_eff = -1.0; int best_flag = -1, best_pref = -1; Shape best = {0, 0, false};
    for (int l2 = 0; l2 <= 6; ++l2) {
        int lpr = 1 << l2;
        int need = (nch + lpr - 1) / lpr;
        int U = 0;
        for (int a : kAllowedU) if (a >= need && a <= max_u) { U = a; break; }
        if (!U) continue;
        double eff = (double)nch / ((double)lpr * U);
        int flag = (lpr >= 8 || lpr >= nch) ? 1 : 0;
        bool better = eff > best_eff + 1e-9 ||
                      (fabs(eff - best_eff) <= 1e-9 && (flag > best_flag || (flag == best_flag && pref[U] > best_pref)));
        if (better) { best_eff = eff; best_flag = flag; best_pref = pref[U]; best.lpr_log2 = l2; best.U = U; }
    }
    if (best.U == 0 || env_int("VG_FORCE_LONG", 0)) { best.lpr_log2 = 6; best.U = VG_LONG_U; best.long_rows = true; }
    int fl = env_int("VG_LPR_LOG2", -1), fu = env_int("VG_U", -1);   // experiment overrides
    if (!best.long_rows && fl >= 0 && fu > 0 && (nch + (1 << fl) - 1) / (1 << fl) <= fu) { best.lpr_log2 = fl; best.U = fu; }
    *out = best;
    return true;
}

template <int VT, int ACC, bool NT>
static scan_fn_t pick_u(int U) {
    switch (U) {
        case 1: return vg_scan_kernel<VT, ACC, 1, NT>;
        case 2: return vg_scan_kernel<VT, ACC, 2, NT>;
        case 3: return vg_scan_kernel<VT, ACC, 3, NT>;
        case 4: return vg_scan_kernel<VT, ACC, 4, NT>;
        case 6: return vg_scan_kernel<VT, ACC, 6, NT>;
        case 8: return vg_scan_kernel<VT, ACC, 8, NT>;
    }
    return nullptr;
}

template <int VT, bool NT>
static scan_fn_t pick_acc(int acc, int U) {
    switch (acc) {
        case A_L2: return pick_u<VT, A_L2, NT>(U);
        case A_COS: return pick_u<VT, A_COS, NT>(U);
        case A_DOT: return pick_u<VT, A_DOT, NT>(U);
        case A_L1: return pick_u<VT, A_L1, NT>(U);
        case A_COSN:
            if constexpr (VT == T_F16 || VT == T_BF16) return pick_u<VT, A_COSN, NT>(U);
            return nullptr;
    }
    return nullptr;
}

template <bool NT>
static scan_fn_t pick_type(int vtype, int acc, int U) {
    switch (vtype) {
        case VG_TYPE_F32: return pick_acc<T_F32, NT>(acc, U);
        case VG_TYPE_U8: return pick_acc<T_U8, NT>(acc, U);
        case VG_TYPE_I8: return pick_acc<T_I8, NT>(acc, U);
        case VG_TYPE_F16: return pick_acc<T_F16, NT>(acc, U);
        case VG_TYPE_BF16: return pick_acc<T_BF16, NT>(acc, U);
    }
    return nullptr;
}

template <int VT, bool NT>
static scan_fn_t pick_long_acc(int acc) {
    switch (acc) {
        case A_L2: return vg_scan_long_kernel<VT, A_L2, NT>;
        case A_COS: return vg_scan_long_kernel<VT, A_COS, NT>;
        case A_DOT: return vg_scan_long_kernel<VT, A_DOT, NT>;
        case A_L1: return vg_scan_long_kernel<VT, A_L1, NT>;
    }
    return nullptr;
}

template <bool NT>
static scan_fn_t pick_long_type(int vtype, int acc) {
    switch (vtype) {
        case VG_TYPE_F32: return pick_long_acc<T_F32, NT>(acc);
        case VG_TYPE_U8: return pick_long_acc<T_U8, NT>(acc);
        case VG_TYPE_I8: return pick_long_acc<T_I8, NT>(acc);
        case VG_TYPE_F16: return pick_long_acc<T_F16, NT>(acc);
        case VG_TYPE_BF16: return pick_long_acc<T_BF16, NT>(acc);
    }
    return nullptr;
}

static scan_fn_t pick_kernel(int vtype, int acc, const Shape &s, bool nt) {
    if (s.long_rows) return nt ? pick_long_type<true>(vtype, acc) : pick_long_type<false>(vtype, acc);
    return nt ? pick_type<true>(vtype, acc, s.U) : pick_type<false>(vtype, acc, s.U);
}

// stream with non-temporal loads once the corpus cannot live in the 256 MiB Infinity Cache anyway
static bool use_nt_loads(const vg_corpus *c) {
    int force = env_int("VG_NT", -1);
    if (force >= 0) return force != 0;
    return c->n_rows * c->stride > (256ll << 20);
}

static int metric_to_acc(int metric) {
    switch (metric) {
        case VG_DIST_L2: case VG_DIST_SQUARED_L2: return A_L2;
        case VG_DIST_COSINE: return A_COS;
        case VG_DIST_DOT: return A_DOT;
        case VG_DIST_L1: return A_L1;
    }
    return -1;
}

static const char *type_tag(int t) {
    switch (t) { case VG_TYPE_F32: return "f32"; case VG_TYPE_F16: return "f16"; case VG_TYPE_BF16: return "bf16";
                 case VG_TYPE_U8: return "u8"; case VG_TYPE_I8: return "i8"; }
    return "?";
}
static const char *acc_tag(int a) {
    switch (a) { case A_L2: return "l2"; case A_COS: return "cos"; case A_DOT: return "dot"; case A_L1: return "l1"; case A_COSN: return "cosn"; }
    return "?";
}

extern "C" const char *vg_scan_kernel_name(vg_corpus *c, int metric) {
    if (!c) return "";
    Shape s;
    int acc = metric_to_acc(metric);
    if (acc < 0 || !choose_shape(c->nch, c->vtype, acc, &s)) return "";
    if (acc == A_COS && (c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16) && !s.long_rows && env_int("VG_HALF_COSN", 1)) acc = A_COSN;
    snprintf(c->kernel_name, sizeof(c->kernel_name), "scan%s_%s_%s_u%d_lpr%d%s", s.long_rows ? "_long" : "",
             type_tag(c->vtype), acc_tag(acc), s.U, 1 << s.lpr_log2, use_nt_loads(c) ? "_nt" : "");
    return c->kernel_name;
}

static int ensure_row_norms(vg_corpus *c);

// Launch the scan (+ merge in top-k mode) on `stream`.  dev_query holds nch*16 zero-padded bytes.
static int launch_scan(vg_corpus *c, int metric, const uint8_t *dev_query, int k, uint64_t *dev_out_keys,
                       float *dev_out_dist, hipStream_t stream) {
    int acc = metric_to_acc(metric);
    if (acc < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    Shape s;
    choose_shape(c->nch, c->vtype, acc, &s);
    // f16 / bf16 cosine: the row norms come from a cached vector (computed once per appended row) instead of being
    // re-accumulated in f64 on every scan - the f64 chain is what bounds these kernels, not HBM
    if (acc == A_COS && (c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16) && !s.long_rows && env_int("VG_HALF_COSN", 1)) {
        int rcn = ensure_row_norms(c);
        if (rcn != VG_OK) return rcn;
        acc = A_COSN;
        if (stream != c->stream) {                       // the norm pass ran on the corpus stream
            if (!c->norm_ev) HIP_TRY(hipEventCreateWithFlags(&c->norm_ev, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c->norm_ev, c->stream));
            HIP_TRY(hipStreamWaitEvent(stream, c->norm_ev, 0));
        }
    }
    scan_fn_t fn = pick_kernel(c->vtype, acc, s, use_nt_loads(c));
    if (!fn) return vg_fail(VG_ERR_UNSUPPORTED, "no scan kernel for type %s", type_tag(c->vtype));

    const int rpb = VG_WAVE >> s.lpr_log2;
    const long long nbatch = (c->n_rows + rpb - 1) / rpb;
    // 16-wave workgroups: one per CU is what ~96 VGPRs admit (5 waves/SIMD); a second one only queues behind it
    const int bpc = std::max(1, std::min(8, env_int("VG_BLOCKS_PER_CU", 1)));
    long long blocks = (nbatch + VG_WAVES_PER_BLOCK - 1) / VG_WAVES_PER_BLOCK;
    blocks = std::max<long long>(1, std::min<long long>(blocks, (long long)c->cu_count * bpc));
    blocks = std::min<long long>(blocks, VG_SEL_MAX_HEADS);          // the final rank-select handles <= 256 lists

    ScanArgs a;
    a.rows = c->d_rows;
    a.query = dev_query;
    a.cand = c->d_cand;
    a.out_dist = dev_out_dist;
    a.n_rows = c->n_rows;
    a.stride = c->stride;
    a.nch = c->nch;
    a.lpr_log2 = s.lpr_log2;
    a.k = k;
    a.root = (metric == VG_DIST_L2) ? 1 : 0;
    a.dim = c->dim;
    a.row_nn = (acc == A_COSN) ? c->d_xnorm : nullptr;
    size_t qbytes = (size_t)c->nch * 16;
    if (s.long_rows) {
        const size_t slice = (size_t)VG_WAVE * VG_LONG_U;               // the long kernel pads the query to whole slices
        qbytes = ((c->nch + slice - 1) / slice) * slice * 16;
    }
    size_t smem = std::max<size_t>(qbytes, (size_t)VG_PUBLISH_LDS_BYTES);
    a.store_lds_off = 0;
    if (dev_out_dist && !s.long_rows) {             // store mode: a staging area per wavefront behind the query
        a.store_lds_off = (int)((qbytes + 255) / 256 * 256);
        smem = std::max<size_t>(smem, (size_t)a.store_lds_off + (size_t)VG_WAVES_PER_BLOCK * VG_STORE_FLOATS * sizeof(float));
    }

    // host appends are only enqueued on the corpus stream: a scan on ANOTHER stream must wait for them
    if (c->append_pending && stream != c->stream) HIP_TRY(hipStreamWaitEvent(stream, c->append_ev, 0));

    hipEvent_t *evs = nullptr;
    if (c->profiling) {
        int slot = (int)(c->prof_launches % VG_PROF_RING);
        evs = &c->ev[(size_t)slot * 3];
        c->ev_had_merge[(size_t)slot] = (dev_out_dist == nullptr);
        ++c->prof_launches;
        hipEventRecord(evs[0], stream);
    }
    if (smem > 64 * 1024)          // very long rows: the query alone needs more than the default dynamic-LDS window
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(VG_BLOCK), smem, stream, a);
    if (evs) hipEventRecord(evs[1], stream);
    if (!dev_out_dist) {
        hipLaunchKernelGGL(vg_merge_kernel, dim3(1), dim3(VG_MERGE_THREADS), 0, stream,
                           (const uint64_t *)c->d_cand, (int)blocks, k, dev_out_keys);
    }
    if (evs) hipEventRecord(evs[2], stream);
    HIP_TRY(hipGetLastError());
    return VG_OK;
}

// latency path for small corpora (see vg_scan_topk): below this size the H2D/D2H staging copies dominate
static bool host_direct(const vg_corpus *c) {
    const int v = env_int("VG_HOST_DIRECT", -1);
    if (v >= 0) return v != 0;
    return c->n_rows * c->stride <= (64ll << 20);
}

static void stage_query(vg_corpus *c, const void *query) {
    memset(c->h_query, 0, (size_t)c->stride);
    memcpy(c->h_query, query, (size_t)c->dim * c->es);
}

// kernel times of ring slot `slot` (waits for that launch to finish)
static void slot_times(vg_corpus *c, int slot, float *scan_ms, float *merge_ms) {
    hipEvent_t *evs = &c->ev[(size_t)slot * 3];
    hipEventSynchronize(evs[2]);
    *scan_ms = 0.f; *merge_ms = 0.f;
    hipEventElapsedTime(scan_ms, evs[0], evs[1]);
    if (c->ev_had_merge[(size_t)slot]) hipEventElapsedTime(merge_ms, evs[1], evs[2]);
}

static void collect_timing(vg_corpus *c) {
    if (!c->profiling || c->prof_launches == 0) return;
    slot_times(c, (int)((c->prof_launches - 1) % VG_PROF_RING), &c->last_scan_ms, &c->last_merge_ms);
}

extern "C" float vg_key_distance(uint64_t key) { return vg_sortable_f32((uint32_t)(key >> 32)); }
extern "C" uint32_t vg_key_position(uint64_t key) { return (uint32_t)(key & 0xFFFFFFFFull); }

extern "C" int vg_scan_topk_device(vg_corpus *c, int metric, const void *dev_query, int k, uint64_t *dev_out_keys,
                                   void *stream) {
    if (!c || !dev_query || !dev_out_keys) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_device: NULL argument");
    if (k < 1 || k > VG_MAX_FUSED_K) return vg_fail(VG_ERR_UNSUPPORTED, "vg_scan_topk_device: k must be in 1..%d", VG_MAX_FUSED_K);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (c->n_rows == 0) {
        HIP_TRY(hipMemsetAsync(dev_out_keys, 0xFF, VG_WAVE * sizeof(uint64_t), st));
        return VG_OK;
    }
    return launch_scan(c, metric, (const uint8_t *)dev_query, k, dev_out_keys, nullptr, st);
}

extern "C" int vg_scan_distances_device(vg_corpus *c, int metric, const void *dev_query, float *dev_out_dist, void *stream) {
    if (!c || !dev_query || !dev_out_dist) return vg_fail(VG_ERR_INVALID, "vg_scan_distances_device: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_rows == 0) return VG_OK;
    return launch_scan(c, metric, (const uint8_t *)dev_query, 0, nullptr, dev_out_dist, stream ? (hipStream_t)stream : c->stream);
}

static int ensure_dist_buffer(vg_corpus *c) {
    if (c->d_dist_cap >= c->n_rows) return VG_OK;
    if (c->d_dist) { hipFree(c->d_dist); c->d_dist = nullptr; c->d_dist_cap = 0; }
    HIP_TRY(hipMalloc(&c->d_dist, (size_t)c->n_rows * sizeof(float)));
    c->d_dist_cap = c->n_rows;
    return VG_OK;
}

extern "C" int vg_scan_distances(vg_corpus *c, int metric, const void *query, float *out_dist_host) {
    if (!c || !query || !out_dist_host) return vg_fail(VG_ERR_INVALID, "vg_scan_distances: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_rows == 0) return VG_OK;
    int rc = ensure_dist_buffer(c);
    if (rc != VG_OK) return rc;
    stage_query(c, query);
    HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
    rc = launch_scan(c, metric, c->d_query, 0, nullptr, c->d_dist, c->stream);
    if (rc != VG_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_dist_host, c->d_dist, (size_t)c->n_rows * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    collect_timing(c);
    return VG_OK;
}

// k > 64 (beyond the fused one-slot-per-lane list): store-mode scan -> key build -> device radix sort
// (vg_select.hip).  Nothing is computed or ordered on the host; k keys come back and are decoded.
extern "C" int vg_select_temp_bytes(long long n, size_t *bytes);
extern "C" int vg_select_sorted_keys(const float *dist, long long n, uint64_t *keys_tmp, uint64_t *keys_sorted,
                                     void *temp, size_t temp_bytes, hipStream_t stream);

extern "C" int vg_select_topk_keys(const float *dist, long long n, uint32_t k, uint64_t *keys_tmp, uint64_t *keys_sorted,
                                   uint32_t cap, void *temp, size_t temp_bytes, uint32_t *state, hipStream_t stream,
                                   uint32_t *out_count);

static int scan_topk_large_k(vg_corpus *c, int metric, const void *query, int k, uint64_t *out_keys, int *out_count) {
    int rc = ensure_dist_buffer(c);
    if (rc != VG_OK) return rc;
    if (c->sel_cap < c->n_rows) {
        if (c->d_sel_keys) hipFree(c->d_sel_keys);
        if (c->d_sel_sorted) hipFree(c->d_sel_sorted);
        if (c->d_sel_temp) hipFree(c->d_sel_temp);
        c->d_sel_keys = c->d_sel_sorted = nullptr; c->d_sel_temp = nullptr; c->sel_cap = 0;
        if (vg_select_temp_bytes(c->n_rows, &c->sel_temp_bytes) != 0) return vg_fail(VG_ERR_HIP, "radix sort temp-size query failed");
        HIP_TRY(hipMalloc(&c->d_sel_keys, (size_t)c->n_rows * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&c->d_sel_sorted, (size_t)c->n_rows * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&c->d_sel_temp, c->sel_temp_bytes ? c->sel_temp_bytes : 16));
        c->sel_cap = c->n_rows;
    }
    stage_query(c, query);
    HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
    rc = launch_scan(c, metric, c->d_query, 0, nullptr, c->d_dist, c->stream);
    if (rc != VG_OK) return rc;
    size_t take = (size_t)std::min<int64_t>((int64_t)k, c->n_rows);
    // radix select (three histogram passes + gather + a sort of ~k keys); the full N-key sort only when the k-th
    // distance has so many ties that the gathered set would not fit
    int sel = 1;
    if (env_int("VG_RADIX_SELECT", 1)) {
        if (!c->d_sel_state) HIP_TRY(hipMalloc(&c->d_sel_state, (4 + 2048) * sizeof(uint32_t)));
        uint32_t got = 0;
        sel = vg_select_topk_keys(c->d_dist, c->n_rows, (uint32_t)take, c->d_sel_keys, c->d_sel_sorted, (uint32_t)std::min<int64_t>(c->sel_cap, 0xFFFFFFFFll),
                                  c->d_sel_temp, c->sel_temp_bytes, c->d_sel_state, c->stream, &got);
        if (sel > 1) return vg_fail(VG_ERR_HIP, "device radix select failed: %s", hipGetErrorString((hipError_t)sel));
        if (sel == 0) take = got;
    }
    if (sel != 0 && vg_select_sorted_keys(c->d_dist, c->n_rows, c->d_sel_keys, c->d_sel_sorted, c->d_sel_temp, c->sel_temp_bytes, c->stream) != 0)
        return vg_fail(VG_ERR_HIP, "device key sort failed: %s", hipGetErrorString(hipGetLastError()));
    if (take) HIP_TRY(hipMemcpyAsync(out_keys, c->d_sel_sorted, take * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    collect_timing(c);
    int cnt = 0;
    while ((size_t)cnt < take && out_keys[cnt] != VG_EMPTY_KEY) ++cnt;     // NaN / +Inf rows sort last and are not results
    *out_count = cnt;
    return VG_OK;
}

// fused path (k <= 64), split in two so that a caller can put several shards in flight before waiting for any:
// enqueue = stage the query + launch scan and merge (+ copy the 64 keys back); collect = wait + hand out the keys
extern "C" int vg_scan_topk_enqueue(vg_corpus *c, int metric, const void *query, int k) {
    if (!c || !query) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_enqueue: NULL argument");
    if (k < 1 || k > VG_MAX_FUSED_K) return vg_fail(VG_ERR_UNSUPPORTED, "vg_scan_topk_enqueue: k must be in 1..%d", VG_MAX_FUSED_K);
    if (metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    c->enqueued = false;
    if (c->n_rows == 0) return VG_OK;
    HIP_TRY(hipSetDevice(c->device));
    stage_query(c, query);
    int rc;
    if (host_direct(c)) {
        // small corpus: the two staging copies cost more than the scan.  h_query / h_keys are pinned, device-mapped
        // host buffers: the kernels read the query and write the k winners straight across the host link.
        rc = launch_scan(c, metric, c->h_query, k, c->h_keys, nullptr, c->stream);
        if (rc != VG_OK) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
        rc = launch_scan(c, metric, c->d_query, k, c->d_keys, nullptr, c->stream);
        if (rc != VG_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->h_keys, c->d_keys, VG_WAVE * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    }
    c->enqueued = true;
    return VG_OK;
}

extern "C" int vg_scan_topk_collect(vg_corpus *c, uint64_t *out_keys64) {
    if (!c || !out_keys64) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_collect: NULL argument");
    if (!c->enqueued) {                                        // empty corpus (or nothing enqueued): no candidates
        for (int i = 0; i < VG_WAVE; ++i) out_keys64[i] = VG_EMPTY_KEY;
        return VG_OK;
    }
    c->enqueued = false;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    collect_timing(c);
    memcpy(out_keys64, c->h_keys, VG_WAVE * sizeof(uint64_t));
    return VG_OK;
}

extern "C" int vg_scan_topk_keys(vg_corpus *c, int metric, const void *query, int k, uint64_t *out_keys, int *out_count) {
    if (!c || !query || !out_count) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_keys: NULL argument");
    *out_count = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_keys) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_keys: NULL output");
    if (metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    HIP_TRY(hipSetDevice(c->device));
    if (k > VG_MAX_FUSED_K) return scan_topk_large_k(c, metric, query, k, out_keys, out_count);
    uint64_t keys[VG_WAVE];
    int rc = vg_scan_topk_enqueue(c, metric, query, k);
    if (rc == VG_OK) rc = vg_scan_topk_collect(c, keys);
    if (rc != VG_OK) return rc;
    int cnt = 0;
    while (cnt < k && keys[cnt] != VG_EMPTY_KEY) { out_keys[cnt] = keys[cnt]; ++cnt; }
    *out_count = cnt;
    return VG_OK;
}

extern "C" int vg_scan_topk(vg_corpus *c, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist,
                            int *out_count) {
    if (!c || !query || !out_count) return vg_fail(VG_ERR_INVALID, "vg_scan_topk: NULL argument");
    *out_count = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_rowids || !out_dist) return vg_fail(VG_ERR_INVALID, "vg_scan_topk: NULL output");
    std::vector<uint64_t> keys((size_t)std::min<int64_t>((int64_t)k, c->n_rows));
    int cnt = 0;
    int rc = vg_scan_topk_keys(c, metric, query, (int)keys.size(), keys.data(), &cnt);
    if (rc != VG_OK) return rc;
    for (int i = 0; i < cnt; ++i) {
        out_dist[i] = (double)vg_key_distance(keys[(size_t)i]);
        out_rowids[i] = vg_corpus_rowid_at(c, (int64_t)vg_key_position(keys[(size_t)i]));
    }
    *out_count = cnt;
    return VG_OK;
}

// ---- batched queries: the MFMA path (vg_batch.hip) when the shape allows it, otherwise nq single-query scans
extern "C" size_t vg_batch_lds_bytes(long long stride_bytes, int k);
extern "C" int vg_batch_launch(const float *dev_rows, long long n_rows, long long stride_bytes,
                               const float *dev_queries, int nq_pad, int nq_real, int k, int mode, int root,
                               const float *dev_xnorm, uint64_t *dev_cand, int npart, int tiles_per_part,
                               uint64_t *dev_out_keys, hipStream_t stream);
extern "C" int vg_batch_lists_per_query(long long n_rows, int npart);
extern "C" int vg_rownorm_launch(const float *dev_rows, long long row0, long long n, long long stride_bytes, float *dev_out,
                                 hipStream_t stream);

// Row norms, computed once per appended row and kept next to the corpus.  f32 corpora: ||row|| (batched cosine / L2);
// f16 / bf16 corpora: (float) sum x^2 (single-query cosine, A_COSN).
static int ensure_row_norms(vg_corpus *c) {
    if (c->xnorm_cap < c->n_rows) {
        float *nb = nullptr;
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipMalloc(&nb, (size_t)cap * sizeof(float)));
        if (c->d_xnorm && c->xnorm_rows > 0)
            HIP_TRY(hipMemcpyAsync(nb, c->d_xnorm, (size_t)c->xnorm_rows * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        if (c->d_xnorm) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->d_xnorm); }
        c->d_xnorm = nb;
        c->xnorm_cap = cap;
    }
    if (c->xnorm_rows < c->n_rows) {
        const long long n = c->n_rows - c->xnorm_rows;
        int rc = 0;
        if (c->vtype == VG_TYPE_F32) {
            rc = vg_rownorm_launch((const float *)c->d_rows, c->xnorm_rows, n, c->stride, c->d_xnorm, c->stream);
        } else {                                   // f16 / bf16: (float) sum x^2, vg_half_rownorm_kernel (vg_scan.h)
            long long blocks = std::min<long long>((n * 16 + 255) / 256, 256 * 32);
            if (c->vtype == VG_TYPE_F16)
                hipLaunchKernelGGL((vg_half_rownorm_kernel<T_F16>), dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_rows,
                                   (long long)c->xnorm_rows, n, (long long)c->stride, c->nch, c->d_xnorm);
            else
                hipLaunchKernelGGL((vg_half_rownorm_kernel<T_BF16>), dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_rows,
                                   (long long)c->xnorm_rows, n, (long long)c->stride, c->nch, c->d_xnorm);
            rc = (int)hipGetLastError();
        }
        if (rc != 0) return vg_fail(VG_ERR_HIP, "row-norm pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->xnorm_rows = c->n_rows;
    }
    return VG_OK;
}

// ---- quantized batches on the integer matrix cores (vg_batch_i8.hip)
extern "C" size_t vg_batch_i8_lds_bytes(long long stride_bytes, int k);
extern "C" int vg_batch_i8_queries_per_block(void);
extern "C" int vg_i8_rowstat_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int is_u8,
                                    int32_t *dev_sx, uint32_t *dev_sxx, uint8_t *dev_flipped, hipStream_t stream);
extern "C" int vg_batch_i8_launch(const uint8_t *dev_rows_signed, long long n_rows, long long stride_bytes,
                                  const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root, int is_u8,
                                  const int32_t *dev_sx, const uint32_t *dev_sxx, uint64_t *dev_cand, int npart,
                                  int tiles_per_part, uint64_t *dev_out_keys, hipStream_t stream);

static bool batch_i8_eligible(const vg_corpus *c, int metric, int k) {
    if (env_int("VG_BATCH_MFMA", 1) == 0) return false;
    if (c->vtype != VG_TYPE_U8 && c->vtype != VG_TYPE_I8) return false;
    if (metric == VG_DIST_L1) return false;
    return vg_batch_i8_lds_bytes(c->stride, k) != 0;
}

// row sums (+ the flipped copy for uint8), once per appended row
static int ensure_i8_row_stats(vg_corpus *c) {
    const bool u8 = (c->vtype == VG_TYPE_U8);
    if (c->i8_cap < c->n_rows) {
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_sx) hipFree(c->d_sx);
        if (c->d_sxx) hipFree(c->d_sxx);
        if (c->d_rows_s8) hipFree(c->d_rows_s8);
        c->d_sx = nullptr; c->d_sxx = nullptr; c->d_rows_s8 = nullptr; c->i8_cap = 0; c->i8_rows = 0;
        // + one tile of slack: the batch kernel fetches the sums of a whole 32-row tile, also behind the last row
        HIP_TRY(hipMalloc(&c->d_sx, (size_t)(cap + 64) * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&c->d_sxx, (size_t)(cap + 64) * sizeof(uint32_t)));
        if (u8) HIP_TRY(hipMalloc(&c->d_rows_s8, (size_t)cap * c->stride));
        c->i8_cap = cap;
    }
    if (c->i8_rows < c->n_rows) {
        int rc = vg_i8_rowstat_launch(c->d_rows, c->i8_rows, c->n_rows - c->i8_rows, c->stride, u8 ? 1 : 0, c->d_sx, c->d_sxx,
                                      c->d_rows_s8, c->stream);
        if (rc != 0) return vg_fail(VG_ERR_HIP, "row-statistics pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->i8_rows = c->n_rows;
    }
    return VG_OK;
}

static bool batch_mfma_eligible(const vg_corpus *c, int metric, int k) {
    if (env_int("VG_BATCH_MFMA", 1) == 0) return false;
    if (c->vtype != VG_TYPE_F32) return false;
    if (metric == VG_DIST_L1) return false;                       // no matrix form
    return vg_batch_lds_bytes(c->stride, k) != 0;
}

static int scan_topk_batch_mfma(vg_corpus *c, int metric, const void *queries, int nq, int k, uint64_t *out_keys,
                                int *out_counts) {
    const bool quantized = (c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8);
    const int QPB = quantized ? vg_batch_i8_queries_per_block() : 128;
    const int nq_pad = ((nq + QPB - 1) / QPB) * QPB;
    const int G = nq_pad / QPB;
    // partitions: enough workgroups to cover the chip (G * npart ~ CUs), a multiple of 8 (one per XCD), <= 256
    int npart = std::max(1, c->cu_count / G);
    if (npart >= 8) npart = (npart / 8) * 8;
    npart = std::min(npart, 256);
    const long long ntiles = (c->n_rows + 31) / 32;
    npart = (int)std::min<long long>(npart, ntiles);
    const int tiles_per_part = (int)((ntiles + npart - 1) / npart);

    const size_t qbytes = (size_t)nq_pad * c->stride;
    const size_t candbytes = (size_t)nq_pad * vg_batch_lists_per_query(c->n_rows, npart) * 64 * sizeof(uint64_t);
    const size_t keybytes = (size_t)nq_pad * 64 * sizeof(uint64_t);
    if (c->bq_bytes < qbytes) { if (c->d_bq) hipFree(c->d_bq); c->d_bq = nullptr; c->bq_bytes = 0;
                                HIP_TRY(hipMalloc(&c->d_bq, qbytes)); c->bq_bytes = qbytes; }
    if (c->bcand_bytes < candbytes) { if (c->d_bcand) hipFree(c->d_bcand); c->d_bcand = nullptr; c->bcand_bytes = 0;
                                      HIP_TRY(hipMalloc(&c->d_bcand, candbytes)); c->bcand_bytes = candbytes; }
    if (c->bkeys_bytes < keybytes) { if (c->d_bkeys) hipFree(c->d_bkeys); c->d_bkeys = nullptr; c->bkeys_bytes = 0;
                                     HIP_TRY(hipMalloc(&c->d_bkeys, keybytes)); c->bkeys_bytes = keybytes; }
    // queries: zero-padded rows of the corpus stride, zero rows up to nq_pad
    std::vector<uint8_t> hq(qbytes, 0);
    const size_t row_bytes = (size_t)c->dim * c->es;
    for (int i = 0; i < nq; ++i) memcpy(hq.data() + (size_t)i * c->stride, (const uint8_t *)queries + (size_t)i * row_bytes, row_bytes);
    HIP_TRY(hipMemcpyAsync(c->d_bq, hq.data(), qbytes, hipMemcpyHostToDevice, c->stream));
    if (quantized) {
        int rcn = ensure_i8_row_stats(c);
        if (rcn != VG_OK) return rcn;
    } else if (metric != VG_DIST_DOT) {
        int rcn = ensure_row_norms(c);
        if (rcn != VG_OK) return rcn;
    }

    hipEvent_t *evs = nullptr;
    if (c->profiling) {
        int slot = (int)(c->prof_launches % VG_PROF_RING);
        evs = &c->ev[(size_t)slot * 3];
        c->ev_had_merge[(size_t)slot] = 0;
        ++c->prof_launches;
        hipEventRecord(evs[0], c->stream);
    }
    const int mode = metric == VG_DIST_DOT ? 0 : (metric == VG_DIST_COSINE ? 1 : 2), root = metric == VG_DIST_L2 ? 1 : 0;
    int rc;
    if (quantized)
        rc = vg_batch_i8_launch(c->vtype == VG_TYPE_U8 ? c->d_rows_s8 : c->d_rows, c->n_rows, c->stride, (const uint8_t *)c->d_bq,
                                nq_pad, nq, k, mode, root, c->vtype == VG_TYPE_U8 ? 1 : 0, c->d_sx, c->d_sxx, c->d_bcand, npart,
                                tiles_per_part, c->d_bkeys, c->stream);
    else
        rc = vg_batch_launch((const float *)c->d_rows, c->n_rows, c->stride, (const float *)c->d_bq, nq_pad, nq, k, mode, root,
                             metric == VG_DIST_DOT ? nullptr : c->d_xnorm, c->d_bcand, npart, tiles_per_part, c->d_bkeys,
                             c->stream);
    if (evs) { hipEventRecord(evs[1], c->stream); hipEventRecord(evs[2], c->stream); }
    if (rc == -1) return -1;
    if (rc != 0) return vg_fail(VG_ERR_HIP, "batched scan launch failed: %s", hipGetErrorString((hipError_t)rc));
    std::vector<uint64_t> keys((size_t)nq * 64);
    HIP_TRY(hipMemcpyAsync(keys.data(), c->d_bkeys, (size_t)nq * 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    collect_timing(c);
    for (int i = 0; i < nq; ++i) {
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            uint64_t key = keys[(size_t)i * 64 + j];
            if (key == VG_EMPTY_KEY) break;
            out_keys[(size_t)i * k + cnt] = key;
            ++cnt;
        }
        out_counts[i] = cnt;
    }
    return VG_OK;
}

extern "C" int vg_scan_topk_batch_keys(vg_corpus *c, int metric, const void *queries, int nq, int k, uint64_t *out_keys,
                                       int *out_counts) {
    if (!c || !queries || !out_counts) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_batch_keys: NULL argument");
    if (nq <= 0) return VG_OK;
    if (metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_keys) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_batch_keys: NULL output");
    HIP_TRY(hipSetDevice(c->device));
    if (batch_mfma_eligible(c, metric, k) || batch_i8_eligible(c, metric, k)) {
        // very large batches go through in slices: the per-(query, partition) candidate lists are nq x ~128 x 512 B
        const int slice = std::max(256, env_int("VG_BATCH_SLICE", 4096));
        int rc = VG_OK;
        const size_t qbytes = (size_t)c->dim * c->es;
        for (int q0 = 0; q0 < nq && rc == VG_OK; q0 += slice) {
            const int nqs = std::min(slice, nq - q0);
            rc = scan_topk_batch_mfma(c, metric, (const uint8_t *)queries + (size_t)q0 * qbytes, nqs, k, out_keys + (size_t)q0 * k,
                                      out_counts + q0);
        }
        if (rc != -1) return rc;
        for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    }
    // shapes the matrix-core kernel does not serve (other types / metrics, k > 32, rows > 512 floats):
    // nq passes of the single-query kernel, still entirely on the GPU
    const uint8_t *q = (const uint8_t *)queries;
    for (int i = 0; i < nq; ++i) {
        int rc = vg_scan_topk_keys(c, metric, q + (size_t)i * c->dim * c->es, k, out_keys + (size_t)i * k, out_counts + i);
        if (rc != VG_OK) return rc;
    }
    return VG_OK;
}

extern "C" int vg_scan_topk_batch(vg_corpus *c, int metric, const void *queries, int nq, int k, int64_t *out_rowids,
                                  double *out_dist, int *out_counts) {
    if (!c || !queries || !out_counts) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_batch: NULL argument");
    if (nq <= 0) return VG_OK;
    for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_rowids || !out_dist) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_batch: NULL output");
    std::vector<uint64_t> keys((size_t)nq * k);
    int rc = vg_scan_topk_batch_keys(c, metric, queries, nq, k, keys.data(), out_counts);
    if (rc != VG_OK) return rc;
    for (int i = 0; i < nq; ++i)
        for (int j = 0; j < out_counts[i]; ++j) {
            const uint64_t key = keys[(size_t)i * k + j];
            out_dist[(size_t)i * k + j] = (double)vg_key_distance(key);
            out_rowids[(size_t)i * k + j] = vg_corpus_rowid_at(c, (int64_t)vg_key_position(key));
        }
    return VG_OK;
}

extern "C" int vg_merge_keys(const uint64_t *keys, int n_lists, int list_len, const int64_t *pos_offsets, int k,
                             int64_t *out_global_pos, double *out_dist) {
    if (!keys || n_lists <= 0 || list_len <= 0 || k <= 0) return 0;
    // heads-of-lists merge; lists are ascending.  Tie on distance -> lower list index first, then lower position:
    // for contiguous row-range shards that IS global scan order.
    std::vector<int> head((size_t)n_lists, 0);
    int cnt = 0;
    while (cnt < k) {
        int best = -1;
        uint64_t bk = VG_EMPTY_KEY;
        for (int l = 0; l < n_lists; ++l) {
            if (head[(size_t)l] >= list_len) continue;
            uint64_t key = keys[(size_t)l * list_len + head[(size_t)l]];
            if (key == VG_EMPTY_KEY) continue;
            // compare by distance image only across lists (positions are list-local)
            if (best < 0 || (key >> 32) < (bk >> 32)) { best = l; bk = key; }
        }
        if (best < 0) break;
        out_dist[cnt] = (double)vg_key_distance(bk);
        out_global_pos[cnt] = (pos_offsets ? pos_offsets[best] : 0) + (int64_t)vg_key_position(bk);
        ++cnt;
        ++head[(size_t)best];
    }
    return cnt;
}

// nq queries at once: keys[list][query][list_len] (what an all_gather of every rank's vg_scan_topk_batch_keys output
// looks like) -> out_global_pos / out_dist [nq][k], out_counts [nq]
extern "C" int vg_merge_keys_batch(const uint64_t *keys, int n_lists, int nq, int list_len, const int64_t *pos_offsets,
                                   int k, int64_t *out_global_pos, double *out_dist, int *out_counts) {
    if (!keys || !out_global_pos || !out_dist || !out_counts || n_lists <= 0 || nq <= 0 || list_len <= 0 || k <= 0) return -1;
    std::vector<uint64_t> one((size_t)n_lists * list_len);
    for (int q = 0; q < nq; ++q) {
        for (int l = 0; l < n_lists; ++l)
            memcpy(&one[(size_t)l * list_len], keys + ((size_t)l * nq + q) * list_len, (size_t)list_len * sizeof(uint64_t));
        out_counts[q] = vg_merge_keys(one.data(), n_lists, list_len, pos_offsets, k, out_global_pos + (size_t)q * k, out_dist + (size_t)q * k);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ query quantizer
// Host C, once per query.  Same arithmetic as the reference (sqlite-vector.c:495-757): s = (v - offset) * scale,
// round half away from zero, clamp; f32 sources use the unguarded int conversion (:524-538), the other source
// types go through the NaN/Inf-aware rounding (:495-515).

static inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, out;
    if (exp == 0x1F) out = sign | 0x7F800000u | (man << 13);
    else if (exp) out = sign | ((exp + 112u) << 23) | (man << 13);
    else if (!man) out = sign;
    else { float v = (float)man * 0x1.0p-24f; memcpy(&out, &v, 4); out |= sign; }
    float f; memcpy(&f, &out, 4); return f;
}
static inline float bf16_to_float(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static inline int cvt_trunc_x86(float r) {          // cvttss2si: NaN / out of range -> INT_MIN
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) return (int)0x80000000u;
    return (int)r;
}

extern "C" int vg_quantize_query(int src_type, const void *src, int dim, float scale, float offset, int qtype, void *dst) {
    if (!src || !dst || dim <= 0) return vg_fail(VG_ERR_INVALID, "vg_quantize_query: bad argument");
    if (qtype != VG_QUANT_U8 && qtype != VG_QUANT_S8) return vg_fail(VG_ERR_INVALID, "vg_quantize_query: qtype must be UINT8 or INT8");
    if (!elem_size(src_type)) return vg_fail(VG_ERR_INVALID, "vg_quantize_query: unknown source type");
    for (int i = 0; i < dim; ++i) {
        float v;
        switch (src_type) {
            case VG_TYPE_F32: v = ((const float *)src)[i]; break;
            case VG_TYPE_F16: v = half_to_float(((const uint16_t *)src)[i]); break;
            case VG_TYPE_BF16: v = bf16_to_float(((const uint16_t *)src)[i]); break;
            case VG_TYPE_U8: v = (float)((const uint8_t *)src)[i]; break;
            default: v = (float)((const int8_t *)src)[i]; break;
        }
        float s = (v - offset) * scale;
        float r = s + 0.5f * (1.0f - 2.0f * (s < 0.0f));
        if (src_type == VG_TYPE_F32) {
            int ir = cvt_trunc_x86(r);
            if (qtype == VG_QUANT_U8) ((uint8_t *)dst)[i] = (uint8_t)(ir > 255 ? 255 : (ir < 0 ? 0 : ir));
            else ((int8_t *)dst)[i] = (int8_t)(ir > 127 ? 127 : (ir < -128 ? -128 : ir));
        } else if (qtype == VG_QUANT_U8) {
            uint8_t o;
            if (!std::isfinite(s)) o = (s > 0.0f) ? 255u : 0u;
            else if (r >= 255.0f) o = 255u;
            else if (r <= 0.0f) o = 0u;
            else o = (uint8_t)(int)r;
            ((uint8_t *)dst)[i] = o;
        } else {
            int8_t o;
            if (!std::isfinite(s)) o = (s > 0.0f) ? 127 : (s < 0.0f ? -128 : 0);
            else if (r >= 127.0f) o = 127;
            else if (r <= -128.0f) o = -128;
            else o = (int8_t)(int)r;
            ((int8_t *)dst)[i] = o;
        }
    }
    return VG_OK;
}

// ------------------------------------------------------------------------------------------------ corpus quantization
// vector_quantize on the staged corpus (vg_quant.hip): min/max pass, then quantize pieces back to the host.

extern "C" int vg_quant_minmax_launch(const uint8_t *rows, long long n_rows, long long stride, int dim, int vtype,
                                      uint32_t *dev_out3, hipStream_t stream);
extern "C" int vg_quant_quantize_launch(const uint8_t *rows, long long row0, long long n_rows, long long stride, int dim,
                                        int vtype, float scale, float offset, int qtype_u8, uint8_t *dev_out,
                                        hipStream_t stream);

extern "C" int vg_corpus_minmax(vg_corpus *c, float *out_min, float *out_max, int *out_any_negative) {
    if (!c || !out_min || !out_max || !out_any_negative) return vg_fail(VG_ERR_INVALID, "vg_corpus_minmax: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t h[3] = {vg_f32_sortable(3.402823466e+38f), vg_f32_sortable(-3.402823466e+38f), 0u};
    if (c->n_rows > 0) {
        uint32_t *d3 = nullptr;
        HIP_TRY(hipMalloc(&d3, sizeof(h)));
        int rc = vg_quant_minmax_launch(c->d_rows, c->n_rows, c->stride, c->dim, c->vtype, d3, c->stream);
        hipError_t e = (rc == 0) ? hipMemcpyAsync(h, d3, sizeof(h), hipMemcpyDeviceToHost, c->stream) : (hipError_t)rc;
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(d3);
        if (e != hipSuccess) return vg_fail(VG_ERR_HIP, "min/max pass failed: %s", hipGetErrorString(e));
    }
    *out_min = vg_sortable_f32(h[0]);
    *out_max = vg_sortable_f32(h[1]);
    *out_any_negative = (int)h[2];
    return VG_OK;
}

extern "C" int vg_corpus_quantize_rows(vg_corpus *c, float scale, float offset, int qtype, int64_t row0, int64_t n_rows,
                                       uint8_t *out_host) {
    if (!c || !out_host) return vg_fail(VG_ERR_INVALID, "vg_corpus_quantize_rows: NULL argument");
    if (qtype != VG_QUANT_U8 && qtype != VG_QUANT_S8) return vg_fail(VG_ERR_INVALID, "vg_corpus_quantize_rows: qtype must be UINT8 or INT8");
    if (row0 < 0 || n_rows < 0 || row0 + n_rows > c->n_rows) return vg_fail(VG_ERR_INVALID, "vg_corpus_quantize_rows: row range out of bounds");
    if (n_rows == 0) return VG_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t piece = std::max<int64_t>(1, (256ll << 20) / c->dim);
    uint8_t *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, (size_t)(std::min(piece, n_rows) * c->dim)));
    for (int64_t r = 0; r < n_rows; r += piece) {
        const int64_t nr = std::min(piece, n_rows - r);
        int rc = vg_quant_quantize_launch(c->d_rows, row0 + r, nr, c->stride, c->dim, c->vtype, scale, offset,
                                          qtype == VG_QUANT_U8 ? 1 : 0, d_out, c->stream);
        hipError_t e = (rc == 0) ? hipMemcpyAsync(out_host + r * c->dim, d_out, (size_t)(nr * c->dim), hipMemcpyDeviceToHost, c->stream)
                                 : (hipError_t)rc;
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { hipFree(d_out); return vg_fail(VG_ERR_HIP, "quantize pass failed: %s", hipGetErrorString(e)); }
    }
    hipFree(d_out);
    return VG_OK;
}

// ------------------------------------------------------------------------------------------------ instrumentation

extern "C" int vg_set_profiling(vg_corpus *c, int enabled) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (enabled && c->ev.empty()) {
        c->ev.assign((size_t)VG_PROF_RING * 3, nullptr);
        c->ev_had_merge.assign((size_t)VG_PROF_RING, 0);
        for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    }
    c->profiling = enabled != 0;
    c->prof_launches = 0;
    return VG_OK;
}

extern "C" int vg_profile_mean_ms(vg_corpus *c, int *n_launches, float *scan_ms, float *merge_ms) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    long long n = std::min<long long>(c->prof_launches, VG_PROF_RING);
    double s = 0.0, m = 0.0;
    for (long long i = 0; i < n; ++i) {
        float a = 0.f, b = 0.f;
        slot_times(c, (int)((c->prof_launches - 1 - i) % VG_PROF_RING), &a, &b);
        s += a; m += b;
    }
    if (n_launches) *n_launches = (int)n;
    if (scan_ms) *scan_ms = n ? (float)(s / n) : 0.f;
    if (merge_ms) *merge_ms = n ? (float)(m / n) : 0.f;
    return VG_OK;
}

extern "C" int vg_last_kernel_ms(vg_corpus *c, float *scan_ms, float *merge_ms) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    collect_timing(c);      // waits for the recorded events of the last (possibly still running) scan
    if (scan_ms) *scan_ms = c->last_scan_ms;
    if (merge_ms) *merge_ms = c->last_merge_ms;
    return VG_OK;
}
