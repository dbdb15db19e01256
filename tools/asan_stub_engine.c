/* stub_engine.c - TEST INFRASTRUCTURE, never shipped and never loaded by the product: a host-memory stand-in for libvectorgpu.so that
 * exports the entry points the extension binds (vext_gpulib.inc), so that the extension's HOST code - staging loops, the parallel
 * reader threads, freshness stamps, vector_quantize's transaction handling - can run under AddressSanitizer / ThreadSanitizer on a box
 * without a GPU (tools/asan_host_check.sh).  f32 rows only, L2 only, no sharding, linear scans: it computes nothing the tests rely on
 * for parity - those go through the real library on the GPU.
 *     gcc -O1 -g -fPIC -shared -o stub.so tools/stub_engine.c -lm          VECTORGPU_LIB=stub.so */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct vg_shards {
    int vtype, dim;
    int64_t n, cap;
    float *rows;
    int64_t *ids;
} vg_shards;

static char g_err[256] = "ok";
static int fail(const char *m) { snprintf(g_err, sizeof(g_err), "%s", m); return -1; }

int vg_device_count(void) { return 1; }
const char *vg_backend_name(void) { return "stub engine (host memory, test infrastructure)"; }
const char *vg_last_error(void) { return g_err; }

int vg_shards_create(const int *devs, int ndev, int vtype, int dim, int64_t block_rows, vg_shards **out) {
    (void)devs; (void)ndev; (void)block_rows;
    if (vtype != 1) return fail("stub engine: f32 only");
    vg_shards *s = (vg_shards *)calloc(1, sizeof(*s));
    if (!s) return fail("out of memory");
    s->vtype = vtype; s->dim = dim;
    *out = s;
    return 0;
}
void vg_shards_destroy(vg_shards *s) { if (s) { free(s->rows); free(s->ids); free(s); } }
int vg_shards_clear(vg_shards *s) { s->n = 0; return 0; }
static int grow(vg_shards *s, int64_t need) {
    if (need <= s->cap) return 0;
    int64_t cap = s->cap ? s->cap : 1024;
    while (cap < need) cap *= 2;
    float *r = (float *)realloc(s->rows, (size_t)cap * s->dim * sizeof(float));
    if (!r) return fail("out of memory");
    s->rows = r;
    int64_t *i = (int64_t *)realloc(s->ids, (size_t)cap * sizeof(int64_t));
    if (!i) return fail("out of memory");
    s->ids = i; s->cap = cap;
    return 0;
}
int vg_shards_reserve(vg_shards *s, int64_t rows) { return grow(s, rows); }
int64_t vg_shards_rows(const vg_shards *s) { return s->n; }
int vg_shards_append(vg_shards *s, const void *rows, int64_t n, int64_t stride, const int64_t *ids) {
    if (grow(s, s->n + n)) return -1;
    for (int64_t i = 0; i < n; ++i) {
        memcpy(s->rows + (s->n + i) * s->dim, (const uint8_t *)rows + i * stride, (size_t)s->dim * sizeof(float));
        s->ids[s->n + i] = ids ? ids[i] : s->n + i + 1;
    }
    s->n += n;
    return 0;
}
int vg_shards_append_records(vg_shards *s, const void *rec, int64_t n) { (void)s; (void)rec; (void)n; return fail("stub engine: no quantized records"); }
int64_t vg_shards_rowid_at(const vg_shards *s, int64_t pos) { return (pos >= 0 && pos < s->n) ? s->ids[pos] : 0; }
int vg_shards_rowids(const vg_shards *s, int64_t pos0, int64_t n, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = vg_shards_rowid_at(s, pos0 + i);
    return 0;
}
int64_t vg_shards_find_rowid(const vg_shards *s, int64_t rowid) {
    for (int64_t i = 0; i < s->n; ++i) if (s->ids[i] == rowid) return i;
    return -1;
}
int vg_shards_scan_distances(vg_shards *s, int metric, const void *q, float *out) {
    (void)metric;
    const float *qf = (const float *)q;
    for (int64_t r = 0; r < s->n; ++r) {
        float acc = 0.0f;
        for (int e = 0; e < s->dim; ++e) { const float d = qf[e] - s->rows[r * s->dim + e]; acc += d * d; }
        out[r] = sqrtf(acc);
    }
    return 0;
}
int vg_shards_scan_topk(vg_shards *s, int metric, const void *q, int k, int64_t *out_ids, double *out_d, int *out_n) {
    float *d = (float *)malloc((size_t)(s->n > 0 ? s->n : 1) * sizeof(float));
    if (!d) return fail("out of memory");
    vg_shards_scan_distances(s, metric, q, d);
    int cnt = 0;
    for (int j = 0; j < k && j < s->n; ++j) {                       /* selection: (distance, position) order */
        int64_t best = -1;
        for (int64_t r = 0; r < s->n; ++r) if (d[r] >= 0.0f && (best < 0 || d[r] < d[best])) best = r;
        if (best < 0) break;
        out_ids[cnt] = s->ids[best]; out_d[cnt] = (double)d[best]; ++cnt;
        d[best] = -1.0f;
    }
    free(d);
    *out_n = cnt;
    return 0;
}
int vg_shards_scan_topk_batch(vg_shards *s, int metric, const void *qs, int nq, int k, int64_t *out_ids, double *out_d, int *out_n) {
    for (int i = 0; i < nq; ++i)
        if (vg_shards_scan_topk(s, metric, (const float *)qs + (size_t)i * s->dim, k, out_ids + (size_t)i * k, out_d + (size_t)i * k, out_n + i)) return -1;
    return 0;
}
int vg_quantize_query(int vtype, const void *src, int dim, float scale, float offset, int qtype, void *out) {
    (void)vtype; (void)src; (void)dim; (void)scale; (void)offset; (void)qtype; (void)out;
    return fail("stub engine: no query quantizer");
}
int vg_shards_minmax(vg_shards *s, float *lo, float *hi, int *has_nan) {
    float a = INFINITY, b = -INFINITY;
    for (int64_t i = 0; i < s->n * s->dim; ++i) { const float v = s->rows[i]; if (v < a) a = v; if (v > b) b = v; }
    *lo = a; *hi = b; if (has_nan) *has_nan = 0;
    return 0;
}
int vg_shards_quantize_rows(vg_shards *s, float scale, float offset, int qtype, int64_t row0, int64_t n, uint8_t *out) {
    for (int64_t r = 0; r < n; ++r)
        for (int e = 0; e < s->dim; ++e) {
            const float v = (s->rows[(row0 + r) * s->dim + e] - offset) * scale;
            int q = (int)(v + (v < 0 ? -0.5f : 0.5f));
            if (qtype == 1) { if (q < 0) q = 0; if (q > 255) q = 255; } else { if (q < -128) q = -128; if (q > 127) q = 127; }
            out[r * s->dim + e] = (uint8_t)q;
        }
    return 0;
}
int vg_shards_set_tie_order(vg_shards *s, int m) { (void)s; (void)m; return 0; }
int vg_shards_set_scan_filter(vg_shards *s, int m) { (void)s; (void)m; return 0; }
int vg_shards_patch_rows(vg_shards *s, const int64_t *pos, int64_t n, const void *rows, int64_t stride) {
    for (int64_t i = 0; i < n; ++i) memcpy(s->rows + pos[i] * s->dim, (const uint8_t *)rows + i * stride, (size_t)s->dim * sizeof(float));
    return 0;
}
int vg_shards_delete_rows(vg_shards *s, const int64_t *pos, int64_t n) {
    for (int64_t i = n - 1; i >= 0; --i) {                         /* ascending positions: delete from the back */
        const int64_t p = pos[i];
        memmove(s->rows + p * s->dim, s->rows + (p + 1) * s->dim, (size_t)(s->n - p - 1) * s->dim * sizeof(float));
        memmove(s->ids + p, s->ids + p + 1, (size_t)(s->n - p - 1) * sizeof(int64_t));
        --s->n;
    }
    return 0;
}
int vg_shards_device_bytes(const vg_shards *s, long long *out3) { out3[0] = (long long)s->cap * s->dim * 4; out3[1] = 0; out3[2] = 0; return 0; }

/* out-of-core scans (vg_slab_scan_*): the stub keeps the k best (distance, position) so far and, for k = 0, every distance - enough for the
 * extension's slab-feeding loops (ooc_scan_full, the stream and batch forms, vector_quantize slab by slab) to run under the sanitizers.
 * It reports 1 TiB of free memory: a table goes out of core here only under VECTORGPU_HBM_LIMIT. */
typedef struct vg_slab_scan {
    int dim, k;
    int64_t rowid_base, seen;
    float *q;
    double *best_d; int64_t *best_id; int nbest;
    float *all_d; int64_t *all_id; int64_t all_n, all_cap;
} vg_slab_scan;
void vg_slab_scan_destroy(vg_slab_scan *s) { if (s) { free(s->q); free(s->best_d); free(s->best_id); free(s->all_d); free(s->all_id); free(s); } }
int vg_slab_scan_begin(int device, int vtype, int dim, int metric, const void *q, int k, int tie, int64_t slab_rows, int64_t rowid_base, vg_slab_scan **out) {
    (void)device; (void)metric; (void)tie; (void)slab_rows;
    if (vtype != 1) return fail("stub engine: f32 only");
    vg_slab_scan *s = (vg_slab_scan *)calloc(1, sizeof(*s));
    if (!s) return fail("out of memory");
    s->dim = dim; s->k = k; s->rowid_base = rowid_base;
    s->q = (float *)malloc((size_t)dim * sizeof(float));
    s->best_d = (double *)malloc((size_t)(k > 0 ? k : 1) * sizeof(double));
    s->best_id = (int64_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(int64_t));
    if (!s->q || !s->best_d || !s->best_id) { vg_slab_scan_destroy(s); return fail("out of memory"); }
    memcpy(s->q, q, (size_t)dim * sizeof(float));
    *out = s;
    return 0;
}
int vg_slab_scan_rows(vg_slab_scan *s, const void *rows, int64_t n, int64_t stride, const int64_t *ids) {
    for (int64_t r = 0; r < n; ++r) {
        const float *x = (const float *)((const uint8_t *)rows + r * stride);
        float acc = 0.0f;
        for (int e = 0; e < s->dim; ++e) { const float d = s->q[e] - x[e]; acc += d * d; }
        const double dist = (double)sqrtf(acc);
        const int64_t id = ids ? ids[r] : s->rowid_base + s->seen;
        ++s->seen;
        if (s->k == 0) {
            if (s->all_n == s->all_cap) {
                const int64_t cap = s->all_cap ? 2 * s->all_cap : 4096;
                float *d2 = (float *)realloc(s->all_d, (size_t)cap * sizeof(float));
                if (!d2) return fail("out of memory");
                s->all_d = d2;
                int64_t *i2 = (int64_t *)realloc(s->all_id, (size_t)cap * sizeof(int64_t));
                if (!i2) return fail("out of memory");
                s->all_id = i2; s->all_cap = cap;
            }
            s->all_d[s->all_n] = (float)dist; s->all_id[s->all_n] = id; ++s->all_n;
            continue;
        }
        int at = s->nbest;                                           /* insert behind equal distances: (distance, position) order */
        while (at > 0 && s->best_d[at - 1] > dist) --at;
        if (at >= s->k) continue;
        const int last = s->nbest < s->k ? s->nbest : s->k - 1;
        for (int j = last; j > at; --j) { s->best_d[j] = s->best_d[j - 1]; s->best_id[j] = s->best_id[j - 1]; }
        s->best_d[at] = dist; s->best_id[at] = id;
        if (s->nbest < s->k) ++s->nbest;
    }
    return 0;
}
int vg_slab_scan_records(vg_slab_scan *s, const void *rec, int64_t n) { (void)s; (void)rec; (void)n; return fail("stub engine: no quantized records"); }
int vg_slab_scan_finish(vg_slab_scan *s, int64_t *ids, double *d, int *n) {
    for (int i = 0; i < s->nbest; ++i) { ids[i] = s->best_id[i]; d[i] = s->best_d[i]; }
    *n = s->nbest;
    return 0;
}
int vg_slab_scan_all(vg_slab_scan *s, int64_t *n, const float **d, const int64_t **ids) { *n = s->all_n; *d = s->all_d; *ids = s->all_id; return 0; }
int vg_device_memory(int device, long long *free_bytes, long long *total_bytes) { (void)device; *free_bytes = 1ll << 40; *total_bytes = 1ll << 40; return 0; }

int vg_shards_trim(vg_shards *s) { (void)s; return 0; }
/* pinned host memory of the host-resident tier: plain memory here */
int vg_host_alloc(size_t bytes, void **out) { *out = malloc(bytes ? bytes : 1); return *out ? 0 : 3; }
void vg_host_free(void *p) { free(p); }

int vg_shards_clone(const vg_shards *src, vg_shards **out) {
    vg_shards *s = (vg_shards *)calloc(1, sizeof(*s));
    if (!s) return fail("out of memory");
    s->vtype = src->vtype; s->dim = src->dim;
    if (grow(s, src->n > 0 ? src->n : 1)) { free(s); return -1; }
    memcpy(s->rows, src->rows, (size_t)src->n * src->dim * sizeof(float));
    memcpy(s->ids, src->ids, (size_t)src->n * sizeof(int64_t));
    s->n = src->n;
    *out = s;
    return 0;
}
