// vg_corpus.hip - the corpus object of the C-ABI (include/vectorgpu.h): device memory for the staged rows, the host
// rowid map, the pinned staging pipeline, plus the small host-side helpers of the ABI that need no scan kernel (key
// decoding / merging, the query quantizer, the vector_quantize wrappers, instrumentation read-out).
// No distance is ever computed on the host: if the HIP runtime / a gfx950 device is missing every entry point fails
// with VG_ERR_NO_DEVICE.
#include "vg_internal.h"

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "vg_device.h"

// ------------------------------------------------------------------------------------------------ errors

static thread_local std::string g_err;

int vg_fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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


// ---- the environment switches (vg_switches.h): the one place that calls getenv
struct VgSwitchName { const char *name; int lab; };
static const VgSwitchName vg_switch_names[VGSW_COUNT] = {
    {"VECTORGPU_SHARD_GATHER", 0},
    {"VECTORGPU_SHARD_THREADS", 0},
    {"VG_BATCH_BPC", 1},
    {"VG_BATCH_H_SPLIT", 0},
    {"VG_BATCH_H_WAVES", 0},
    {"VG_BATCH_LONG", 0},
    {"VG_BATCH_LONG_BPC", 1},
    {"VG_BATCH_MFMA", 0},
    {"VG_BATCH_MIN_QUERIES", 1},
    {"VG_BATCH_PREPASS", 1},
    {"VG_BATCH_Q8", 0},
    {"VG_BATCH_SLICE", 0},
    {"VG_BATCH_STAGES", 0},
    {"VG_BATCH_TILE_MAJOR", 0},
    {"VG_BLOCKS_PER_CU", 1},
    {"VG_F32_FILTER", 0},
    {"VG_FILTER_LPR_LOG2", 1},
    {"VG_FILTER_U", 1},
    {"VG_FORCE_LONG", 0},
    {"VG_HALF_COSN", 1},
    {"VG_HOST_DIRECT", 1},
    {"VG_KEYS_DIRECT", 1},
    {"VG_LPR_LOG2", 1},
    {"VG_MULTI_SCAN", 0},
    {"VG_NT", 1},
    {"VG_Q8_TWO_READS", 1},
    {"VG_RADIX_SELECT", 0},
    {"VG_REF_ALWAYS_EMIT", 1},
    {"VG_REF_STORE_MODE", 0},
    {"VG_SCAN_FILTER", 0},
    {"VG_SCAN_FILTER_MIN_MB", 0},
    {"VG_SCAN_FILTER_MIRROR_COPY", 1},
    {"VG_SCAN_FILTER_N4", 0},
    {"VG_SCAN_FILTER_NO_GUARD", 0},
    {"VG_SCAN_FILTER_PREMERGE", 1},
    {"VG_SCAN_FILTER_PREPASS", 1},
    {"VG_SCAN_FILTER_PREPASS_DIV", 1},
    {"VG_SCAN_FILTER_SHADOW", 0},
    {"VG_SCAN_ORDER", 1},
    {"VG_SHAPE_BF16_L2_U3", 1},
    {"VG_SHAPE_F16_ROUND3", 1},
    {"VG_SHAPE_INT_SHORT_ROUND3", 1},
    {"VG_SHAPE_PREF_ROUND1", 1},
    {"VG_U", 1},
};
std::atomic<int> vg_switch_values[VGSW_COUNT];
void vg_switches_read(void) {
    for (int i = 0; i < VGSW_COUNT; ++i) {
        int v = VGSW_UNSET;
#ifndef VG_LAB
        if (!vg_switch_names[i].lab)
#endif
        {
            const char *e = getenv(vg_switch_names[i].name);
            if (e && *e) {
                if (i == SW_VG_SCAN_FILTER_SHADOW) v = (unsigned char)e[0];                     // 'b' / 'r': the bf16 copy / the rows themselves
                else if (i == SW_VECTORGPU_SHARD_GATHER) v = (e[0] == 'r' || e[0] == 'R') ? 1 : 0;                 // "rccl"
                else v = atoi(e);
            }
        }
        vg_switch_values[i].store(v, std::memory_order_relaxed);
    }
}
__attribute__((constructor)) static void vg_switches_at_load(void) { vg_switches_read(); }     // (before the first corpus: the table is never all zeros)
extern "C" void vg_reload_switches(void) { vg_switches_read(); }

extern "C" int vg_corpus_create(int device, int vtype, int dim, int64_t capacity_rows_hint, vg_corpus **out) {
    if (!out) return vg_fail(VG_ERR_INVALID, "vg_corpus_create: out is NULL");
    *out = nullptr;
    vg_switches_read();                                       // a corpus takes the environment's switches as they are NOW (vg_switches.h)
    int es = vg_elem_size(vtype);
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
    if (c->d_cand_pre) hipFree(c->d_cand_pre);
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
    if (c->h_bq) hipHostFree(c->h_bq);
    if (c->d_xnorm) hipFree(c->d_xnorm);
    if (c->d_sel_state) hipFree(c->d_sel_state);
    if (c->d_sx) hipFree(c->d_sx);
    if (c->d_rows_s8) hipFree(c->d_rows_s8);
    if (c->d_rows_tm) hipFree(c->d_rows_tm);
    if (c->d_rows_bf) hipFree(c->d_rows_bf);
    if (c->d_rows_n4) hipFree(c->d_rows_n4);
    if (c->d_n4stat) hipFree(c->d_n4stat);
    if (c->d_rows_q8) hipFree(c->d_rows_q8);
    if (c->d_q8stat) hipFree(c->d_q8stat);
    if (c->d_rows_q8tm) hipFree(c->d_rows_q8tm);
    if (c->d_q8tm_stat) hipFree(c->d_q8tm_stat);
    if (c->d_filter_evals) hipFree(c->d_filter_evals);
    if (c->h_filter_evals) hipHostFree(c->h_filter_evals);
    if (c->d_below) hipFree(c->d_below);
    if (c->d_ref_prefix) hipFree(c->d_ref_prefix);
    if (c->h_ref) hipHostFree(c->h_ref);
    if (c->norm_ev) hipEventDestroy(c->norm_ev);
    if (c->d_bcand) hipFree(c->d_bcand);
    if (c->d_bkeys) hipFree(c->d_bkeys);
    if (c->d_bpairs) hipFree(c->d_bpairs);
    if (c->d_bpcounts) hipFree(c->d_bpcounts);
    for (hipEvent_t e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int vg_corpus_clear(vg_corpus *c) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    c->n_rows = 0;
    c->dist_valid_rows = 0;
    c->xnorm_rows = 0;
    c->i8_rows = 0;
    c->tm_rows = 0;
    c->bf_rows = 0;
    c->q8_rows = 0;
    c->q8tm_rows = 0;
    c->n4_rows = 0;
    c->n4_probe = 0;
    c->rowids.clear();
    c->rowids_ascending = true;
    return VG_OK;
}

extern "C" int64_t vg_corpus_rows(const vg_corpus *c) { return c ? c->n_rows : 0; }
extern "C" int vg_corpus_dim(const vg_corpus *c) { return c ? c->dim : 0; }
extern "C" int vg_corpus_type(const vg_corpus *c) { return c ? c->vtype : 0; }
extern "C" int vg_corpus_device(const vg_corpus *c) { return c ? c->device : -1; }
extern "C" int64_t vg_corpus_hbm_bytes(const vg_corpus *c) { return c ? c->cap_rows * c->stride : 0; }
// what this corpus holds on its device, by allocation (the runtime knows every allocation's size): [0] the row matrix, [1] per-row data
// derived from it - shadow copies of the filter scans, tile-major copies of the batch kernels, norms and row statistics - [2] working
// buffers (queries, candidate lists, distances, pair regions, staging)
extern "C" int vg_corpus_device_bytes(const vg_corpus *c, long long *out3) {
    if (!c || !out3) return vg_fail(VG_ERR_INVALID, "vg_corpus_device_bytes: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    auto size_of = [](const void *p) -> long long {
        if (!p) return 0;
        size_t n = 0;
        if (hipMemPtrGetInfo(const_cast<void *>(p), &n) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return (long long)n;
    };
    out3[0] = size_of(c->d_rows);
    const void *derived[] = {c->d_rows_s8, c->d_sx, c->d_rows_tm, c->d_rows_bf, c->d_rows_q8, c->d_q8stat, c->d_rows_q8tm, c->d_q8tm_stat, c->d_rows_n4, c->d_n4stat, c->d_xnorm};
    const void *working[] = {c->d_query, c->d_cand, c->d_cand_pre, c->d_keys, c->d_dist, c->d_below, c->d_ref_prefix, c->d_sel_keys, c->d_sel_sorted,
                             c->d_sel_temp, c->d_sel_state, c->d_stage, c->d_filter_evals, c->d_bq, c->d_bcand, c->d_bkeys, c->d_bpairs, c->d_bpcounts};
    out3[1] = 0; out3[2] = 0;
    for (const void *p : derived) out3[1] += size_of(p);
    for (const void *p : working) out3[2] += size_of(p);
    return VG_OK;
}

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

// A second corpus with the same rows, rowids and per-corpus switches (device-to-device copy; per-row copies derived from the rows are NOT
// copied - the clone makes its own when a scan wants them).  What the extension's registry of shared copies uses for copy-on-write: a
// connection that has to CHANGE a copy other connections hold (append the rows it inserted, patch the rows it updated) clones it - a few
// milliseconds per 10 GB - instead of reading the table again.  VG_ERR_NOMEM when the device has no room for a second copy.
extern "C" int vg_corpus_clone(const vg_corpus *src, vg_corpus **out) {
    if (!src || !out) return vg_fail(VG_ERR_INVALID, "vg_corpus_clone: NULL argument");
    *out = nullptr;
    vg_corpus *c = nullptr;
    int rc = vg_corpus_create(src->device, src->vtype, src->dim, std::max<int64_t>(src->n_rows, 1024), &c);
    if (rc != VG_OK) return rc;
    if (src->n_rows > 0) {
        hipError_t e = hipStreamSynchronize(src->stream);                                   // (appends of the source still in flight)
        if (e == hipSuccess) e = hipMemcpyAsync(c->d_rows, src->d_rows, (size_t)(src->n_rows * src->stride), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { vg_corpus_destroy(c); return vg_fail(VG_ERR_HIP, "corpus clone copy failed: %s", hipGetErrorString(e)); }
    }
    c->n_rows = src->n_rows;
    c->rowids = src->rowids;
    c->rowids_ascending = src->rowids_ascending;
    c->rowid_base = src->rowid_base;
    c->tie_order = src->tie_order;
    c->scan_filter_mode = src->scan_filter_mode;
    *out = c;
    return VG_OK;
}

// Give back what a too-generous reservation holds beyond the rows that arrived (the extension reserves from a cheap UPPER bound of the
// table's row count - the key span - which sparse keys can put several times above the count; the per-row copies derived later - norms,
// shadow and tile-major copies - are sized by cap_rows too, so the excess would multiply: ADVICE r4).  A no-op within 25 % + 1024 rows.
extern "C" int vg_corpus_trim(vg_corpus *c) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    const int64_t keep = std::max<int64_t>(c->n_rows, 1024);
    if (!c->d_rows || c->cap_rows <= keep + keep / 4 + 1024) return VG_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint8_t *nb = nullptr;
    if (hipMalloc(&nb, (size_t)(keep * c->stride)) != hipSuccess) { (void)hipGetLastError(); return VG_OK; }   // (no room for the smaller copy: keep the large one)
    hipError_t e = hipSuccess;
    if (c->n_rows > 0) e = hipMemcpyAsync(nb, c->d_rows, (size_t)(c->n_rows * c->stride), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { hipFree(nb); return vg_fail(VG_ERR_HIP, "corpus trim copy failed: %s", hipGetErrorString(e)); }
    hipFree(c->d_rows);
    c->d_rows = nb;
    c->cap_rows = keep;
    return VG_OK;
}

extern "C" int vg_corpus_reserve(vg_corpus *c, int64_t capacity_rows) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return corpus_reserve(c, capacity_rows);
}

static void note_rowids(vg_corpus *c, const int64_t *rowids, int64_t n) {
    if (rowids && c->rowids_ascending) {                       // (vg_corpus_find_rowid searches by bisection)
        int64_t prev = c->rowids.empty() ? (c->n_rows > 0 ? c->rowid_base + c->n_rows - 1 : INT64_MIN) : c->rowids.back();
        for (int64_t i = 0; i < n; ++i) {
            if (rowids[i] <= prev && !(i == 0 && prev == INT64_MIN)) { c->rowids_ascending = false; break; }
            prev = rowids[i];
        }
    }
    if (rowids) {
        if (c->rowids.empty() && c->n_rows > 0) {
            c->rowids.resize((size_t)c->n_rows);
            for (int64_t i = 0; i < c->n_rows; ++i) c->rowids[(size_t)i] = c->rowid_base + i;
        }
        c->rowids.insert(c->rowids.end(), rowids, rowids + n);
    } else if (!c->rowids.empty()) {
        // implicit rowids behind explicit ones: base + position may fall below the last explicit rowid - then the map is no longer
        // ascending and vg_corpus_find_rowid must not bisect it
        if (n > 0 && c->rowid_base + c->n_rows <= c->rowids.back()) c->rowids_ascending = false;
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

// the same for rows that go to scattered places (vg_corpus_patch_rows): source row r lands at row positions[r], padded with zeros
__global__ void vg_scatter_rows_kernel(const uint8_t *src, long long src_stride, int row_bytes, const long long *positions,
                                       uint8_t *dst, long long dst_stride, int nch, long long n_rows) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * nch) return;
    const long long r = t / nch;
    const int ch = (int)(t - r * nch);
    const uint8_t *s = src + r * src_stride + (long long)ch * 16;
    const int remain = row_bytes - ch * 16;
    uint32_t w[4] = {0, 0, 0, 0};
    if (remain >= 16 && ((reinterpret_cast<uintptr_t>(s) & 3) == 0)) {
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s);
        w[0] = s4[0]; w[1] = s4[1]; w[2] = s4[2]; w[3] = s4[3];
    } else {
        const int nb = remain < 16 ? remain : 16;
        for (int j = 0; j < nb; ++j) w[j >> 2] |= (uint32_t)s[j] << ((j & 3) * 8);
    }
    *reinterpret_cast<uint4 *>(dst + positions[r] * dst_stride + (long long)ch * 16) = make_uint4(w[0], w[1], w[2], w[3]);
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
// ---- pinned host memory handed out by the engine (round 6: the residency tier between "in HBM" and "re-read the table per query")
// A table that does not fit the device is kept ONCE in pinned host memory by the extension and streamed over PCIe, slab by slab, for every
// query (vext_staging.inc: ooc_scan_full) - ~50 GB/s instead of the 1-8 GB/s of sqlite3_step.  Rows appended from such a block skip the
// bounce buffers: the DMA engine reads the caller's pages directly.
static std::mutex g_host_mu;
static std::vector<std::pair<uintptr_t, size_t>> g_host_blocks;
extern "C" int vg_host_alloc(size_t bytes, void **out) {
    if (!out) return vg_fail(VG_ERR_INVALID, "vg_host_alloc: out is NULL");
    *out = nullptr;
    if (bytes == 0) return vg_fail(VG_ERR_INVALID, "vg_host_alloc: zero bytes");
    if (vg_device_count() <= 0) return vg_fail(VG_ERR_NO_DEVICE, "no HIP device available");
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
        (void)hipGetLastError();
        return vg_fail(VG_ERR_NOMEM, "vg_host_alloc: %zu bytes of pinned host memory refused", bytes);
    }
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_blocks.emplace_back((uintptr_t)p, bytes);
    *out = p;
    return VG_OK;
}
extern "C" void vg_host_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        for (size_t i = 0; i < g_host_blocks.size(); ++i)
            if (g_host_blocks[i].first == (uintptr_t)p) { g_host_blocks.erase(g_host_blocks.begin() + (long)i); break; }
    }
    (void)hipHostFree(p);
}
static bool host_block_holds(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (const auto &b : g_host_blocks)
        if ((uintptr_t)p >= b.first && (uintptr_t)p + bytes <= b.first + b.second) return true;
    return false;
}

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
    if (host_block_holds(src, (size_t)((n_rows - 1) * src_stride + src_off + row_bytes))) {
        // rows in pinned memory of ours (vg_host_alloc): no bounce copy - one DMA for the whole run when the layouts agree, else piece by
        // piece through the device-side repack (stream-ordered: a piece's repack has read d_stage before the next piece's copy lands)
        uint8_t *unused_pin;
        int unused_slot;
        if ((rc = pin_acquire(c, &unused_pin, &unused_slot)) != VG_OK) return rc;           // (creates d_stage and the append event on first use)
        if (same_layout) {
            HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(n_rows * c->stride), hipMemcpyHostToDevice, c->stream));
        } else {
            for (int64_t r0 = 0; r0 < n_rows; r0 += piece_rows) {
                const int64_t nr = std::min(piece_rows, n_rows - r0);
                const size_t bytes = (size_t)((nr - 1) * src_stride + src_off + row_bytes);
                HIP_TRY(hipMemcpyAsync(c->d_stage, (const uint8_t *)src + r0 * src_stride, bytes, hipMemcpyHostToDevice, c->stream));
                long long total = nr * c->nch;
                hipLaunchKernelGGL(vg_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                                   (const uint8_t *)c->d_stage, (long long)src_stride, src_off, (int)row_bytes,
                                   dst + r0 * c->stride, (long long)c->stride, c->nch, (long long)nr);
            }
        }
        HIP_TRY(hipEventRecord(c->append_ev, c->stream));
        c->append_pending = true;
        HIP_TRY(hipGetLastError());
        return VG_OK;
    }
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

// process-wide count of rows that went to a device through any vg_corpus_append* call (tests observe staging with it:
// an append-only re-stage of the extension must move it by the number of new rows, not by the table size)
static std::atomic<long long> g_rows_appended{0};
extern "C" long long vg_stat_rows_appended(void) { return g_rows_appended.load(); }

extern "C" int vg_corpus_append(vg_corpus *c, const void *host_rows, int64_t n_rows, int64_t row_stride_bytes,
                                const int64_t *rowids) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    if (n_rows == 0) return VG_OK;
    if (!host_rows || n_rows < 0) return vg_fail(VG_ERR_INVALID, "vg_corpus_append: bad rows pointer / count");
    if (row_stride_bytes < (int64_t)c->dim * c->es) return vg_fail(VG_ERR_INVALID, "vg_corpus_append: stride %lld smaller than a row (%lld bytes)", (long long)row_stride_bytes, (long long)c->dim * c->es);
    int rc = append_impl(c, host_rows, false, n_rows, row_stride_bytes, 0);
    if (rc != VG_OK) return rc;
    g_rows_appended += n_rows;
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
    g_rows_appended += n_rows;
    note_rowids(c, host_rowids, n_rows);
    c->n_rows += n_rows;
    return VG_OK;
}

// ------------------------------------------------------------------------------------------------ row maintenance
// The reference re-reads the table for every scan (sqlite-vector.c:2077-2107), so an UPDATE or DELETE is visible to the next
// query for free.  A corpus resident in HBM needs the equivalent: overwrite single rows in place, take rows out and close the
// gaps - without a pass over the table.  Everything derived per row (norms, shadow copies, tile-major copies, row sums) is
// re-made from the first touched row on, by the passes that extend it after an append.

static void invalidate_derived_from(vg_corpus *c, int64_t pos) {
    c->xnorm_rows = std::min(c->xnorm_rows, pos);
    c->i8_rows = std::min(c->i8_rows, pos);
    c->tm_rows = std::min(c->tm_rows, pos);
    c->bf_rows = std::min(c->bf_rows, pos);
    c->q8_rows = std::min(c->q8_rows, pos);
    c->q8tm_rows = std::min(c->q8tm_rows, pos);
    c->n4_rows = std::min(c->n4_rows, pos);
    c->dist_valid_rows = 0;
}

// position of `rowid`, -1 if the corpus does not hold it, -2 if its rowids are not ascending (no lookup: the caller re-stages)
extern "C" int64_t vg_corpus_find_rowid(const vg_corpus *c, int64_t rowid) {
    if (!c) return -1;
    if (c->rowids.empty()) {
        const int64_t p = rowid - c->rowid_base;
        return (p >= 0 && p < c->n_rows) ? p : -1;
    }
    if (!c->rowids_ascending) return -2;
    auto it = std::lower_bound(c->rowids.begin(), c->rowids.end(), rowid);
    return (it != c->rowids.end() && *it == rowid) ? (int64_t)(it - c->rowids.begin()) : -1;
}

// rows at `positions` (any order, all DISTINCT) are overwritten with host_rows[i]
extern "C" int vg_corpus_patch_rows(vg_corpus *c, const int64_t *positions, int64_t n, const void *host_rows, int64_t row_stride_bytes) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    if (n == 0) return VG_OK;
    const int64_t row_bytes = (int64_t)c->dim * c->es;
    if (!positions || !host_rows || n < 0 || row_stride_bytes < row_bytes) return vg_fail(VG_ERR_INVALID, "vg_corpus_patch_rows: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (row_stride_bytes > VG_PIN_BYTES) return vg_fail(VG_ERR_UNSUPPORTED, "row stride %lld exceeds the staging buffer", (long long)row_stride_bytes);
    int64_t first = c->n_rows;
    for (int64_t i = 0; i < n; ++i) {
        if (positions[i] < 0 || positions[i] >= c->n_rows) return vg_fail(VG_ERR_INVALID, "vg_corpus_patch_rows: position %lld out of range", (long long)positions[i]);
        first = std::min(first, positions[i]);
    }
    // rows and their positions travel together in one pinned piece; one kernel scatters the piece's rows to their places
    const int64_t piece_rows = std::max<int64_t>(1, (VG_PIN_BYTES - 8) / (row_stride_bytes + 8));
    for (int64_t r0 = 0; r0 < n; r0 += piece_rows) {
        const int64_t nr = std::min(piece_rows, n - r0);
        const size_t bytes = (size_t)((nr - 1) * row_stride_bytes + row_bytes);
        const size_t pos_off = (bytes + 7) & ~(size_t)7;
        uint8_t *pin;
        int slot;
        int rc = pin_acquire(c, &pin, &slot);
        if (rc != VG_OK) return rc;
        memcpy(pin, (const uint8_t *)host_rows + r0 * row_stride_bytes, bytes);
        memcpy(pin + pos_off, positions + r0, (size_t)nr * sizeof(int64_t));
        HIP_TRY(hipMemcpyAsync(c->d_stage, pin, pos_off + (size_t)nr * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        const long long total = nr * c->nch;
        hipLaunchKernelGGL(vg_scatter_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                           (const uint8_t *)c->d_stage, (long long)row_stride_bytes, (int)row_bytes,
                           reinterpret_cast<const long long *>(c->d_stage + pos_off), c->d_rows, (long long)c->stride, c->nch, (long long)nr);
        HIP_TRY(hipEventRecord(c->pin_ev[slot], c->stream));
        c->pin_busy[slot] = true;
    }
    HIP_TRY(hipEventRecord(c->append_ev, c->stream));
    c->append_pending = true;
    HIP_TRY(hipGetLastError());
    invalidate_derived_from(c, first);
    return VG_OK;
}

// rows at `positions` (strictly ascending) leave the corpus; the rows behind them move up, scan order is kept
extern "C" int vg_corpus_delete_rows(vg_corpus *c, const int64_t *positions, int64_t n) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    if (n == 0) return VG_OK;
    if (!positions || n < 0 || n > c->n_rows) return vg_fail(VG_ERR_INVALID, "vg_corpus_delete_rows: bad argument");
    for (int64_t i = 0; i < n; ++i)
        if (positions[i] < 0 || positions[i] >= c->n_rows || (i > 0 && positions[i] <= positions[i - 1]))
            return vg_fail(VG_ERR_INVALID, "vg_corpus_delete_rows: positions must be ascending and inside the corpus");
    HIP_TRY(hipSetDevice(c->device));
    // segment j (rows between deletion j-1 and deletion j) moves down by j rows.  Source and destination overlap when a segment is
    // longer than its shift: such a move goes through a bounce buffer chunk by chunk (ascending, so a chunk's destination has
    // been consumed already); a shift of at least the chunk size copies directly in shift-sized pieces.
    const int64_t tail_rows = c->n_rows - positions[0];
    size_t tmp_bytes = (size_t)std::min<int64_t>(tail_rows * c->stride, 256ll << 20);
    uint8_t *tmp = nullptr;
    if (hipMalloc(&tmp, tmp_bytes) != hipSuccess) {
        (void)hipGetLastError();
        tmp = nullptr;
        uint8_t *pin; int slot;                                  // (allocates the 16 MiB device staging area on first use)
        int rc = pin_acquire(c, &pin, &slot);
        if (rc != VG_OK) return rc;
        tmp_bytes = (size_t)VG_PIN_BYTES;
    }
    uint8_t *bounce = tmp ? tmp : c->d_stage;
    const int64_t bounce_rows = std::max<int64_t>(1, (int64_t)(tmp_bytes / (size_t)c->stride));
    hipError_t e = hipSuccess;
    for (int64_t j = 1; j <= n && e == hipSuccess; ++j) {
        const int64_t from = positions[j - 1] + 1, to = (j < n) ? positions[j] : c->n_rows, shift = j;
        int64_t r = from;
        while (r < to && e == hipSuccess) {
            if (shift >= bounce_rows || to - r <= shift) {       // no overlap inside one piece of at most `shift` rows
                const int64_t m = std::min(shift, to - r);
                e = hipMemcpyAsync(c->d_rows + (r - shift) * c->stride, c->d_rows + r * c->stride, (size_t)(m * c->stride), hipMemcpyDeviceToDevice, c->stream);
                r += m;
            } else {
                const int64_t m = std::min(bounce_rows, to - r);
                e = hipMemcpyAsync(bounce, c->d_rows + r * c->stride, (size_t)(m * c->stride), hipMemcpyDeviceToDevice, c->stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(c->d_rows + (r - shift) * c->stride, bounce, (size_t)(m * c->stride), hipMemcpyDeviceToDevice, c->stream);
                r += m;
            }
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (tmp) hipFree(tmp);
    if (e != hipSuccess) return vg_fail(VG_ERR_HIP, "row compaction failed: %s", hipGetErrorString(e));
    if (!c->rowids.empty()) {
        size_t w = (size_t)positions[0];
        int64_t next = 0;
        for (size_t r = (size_t)positions[0]; r < (size_t)c->n_rows; ++r) {
            if (next < n && (int64_t)r == positions[next]) { ++next; continue; }
            c->rowids[w++] = c->rowids[r];
        }
        c->rowids.resize(w);
    } else {                                                     // implicit rowids stop being base + position
        std::vector<int64_t> ids;
        ids.reserve((size_t)(c->n_rows - n));
        int64_t next = 0;
        for (int64_t r = 0; r < c->n_rows; ++r) {
            if (next < n && r == positions[next]) { ++next; continue; }
            ids.push_back(c->rowid_base + r);
        }
        c->rowids.swap(ids);
    }
    c->n_rows -= n;
    invalidate_derived_from(c, positions[0]);
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
    g_rows_appended += n_records;
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
    if (!vg_elem_size(src_type)) return vg_fail(VG_ERR_INVALID, "vg_quantize_query: unknown source type");
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
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipEventCreate(&e0); hipEventCreate(&e1);
        HIP_TRY(hipMemcpyAsync(d3, h, sizeof(h), hipMemcpyHostToDevice, c->stream));     // (so that the events bracket the kernel alone)
        hipEventRecord(e0, c->stream);
        int rc = vg_quant_minmax_launch(c->d_rows, c->n_rows, c->stride, c->dim, c->vtype, d3, c->stream);
        hipEventRecord(e1, c->stream);
        hipError_t e = (rc == 0) ? hipMemcpyAsync(h, d3, sizeof(h), hipMemcpyDeviceToHost, c->stream) : (hipError_t)rc;
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) { hipEventElapsedTime(&c->pass_ms[0], e0, e1); c->pass_rows[0] = c->n_rows; }
        hipEventDestroy(e0); hipEventDestroy(e1);
        hipFree(d3);
        if (e != hipSuccess) return vg_fail(VG_ERR_HIP, "min/max pass failed: %s", hipGetErrorString(e));
    }
    *out_min = vg_sortable_f32(h[0]);
    *out_max = vg_sortable_f32(h[1]);
    *out_any_negative = (int)h[2];
    return VG_OK;
}

// kernel milliseconds + rows of the last minmax (0) / quantize (1) / int8-shadow (2) pass of this corpus
extern "C" int vg_corpus_pass_ms(const vg_corpus *c, int which, float *out_ms, long long *out_rows) {
    if (!c || which < 0 || which > 2 || !out_ms || !out_rows) return vg_fail(VG_ERR_INVALID, "vg_corpus_pass_ms: bad argument");
    *out_ms = c->pass_ms[which];
    *out_rows = c->pass_rows[which];
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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEventCreate(&e0); hipEventCreate(&e1);
    c->pass_ms[1] = 0.f; c->pass_rows[1] = 0;
    for (int64_t r = 0; r < n_rows; r += piece) {
        const int64_t nr = std::min(piece, n_rows - r);
        hipEventRecord(e0, c->stream);
        int rc = vg_quant_quantize_launch(c->d_rows, row0 + r, nr, c->stride, c->dim, c->vtype, scale, offset,
                                          qtype == VG_QUANT_U8 ? 1 : 0, d_out, c->stream);
        hipEventRecord(e1, c->stream);
        hipError_t e = (rc == 0) ? hipMemcpyAsync(out_host + r * c->dim, d_out, (size_t)(nr * c->dim), hipMemcpyDeviceToHost, c->stream)
                                 : (hipError_t)rc;
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d_out); return vg_fail(VG_ERR_HIP, "quantize pass failed: %s", hipGetErrorString(e)); }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        c->pass_ms[1] += ms; c->pass_rows[1] += nr;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(d_out);
    return VG_OK;
}
