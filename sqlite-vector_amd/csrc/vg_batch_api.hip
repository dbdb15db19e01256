// vg_batch_api.hip - host side of the batched scans (vg_scan_topk_batch[_keys]): partition planning, the cached per-row
// statistics the matrix-core kernels need, query staging, launches of vg_batch.hip (f32) / vg_batch_i8.hip (uint8, int8) /
// vg_batch_h.hip (f16, bf16), slicing of very large batches; shapes none of them serves go through the multi-query scan
// (vg_multi.hip) or, last, one single-query scan per query.
#include "vg_internal.h"

#include "vg_device.h"

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
// ---- quantized batches on the integer matrix cores (vg_batch_i8.hip)
extern "C" size_t vg_batch_i8_lds_bytes(long long stride_bytes, int k);
extern "C" int vg_batch_i8_queries_per_block(long long stride_bytes);
extern "C" int vg_i8_rowstat_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int is_u8,
                                    uint32_t *dev_stat, uint8_t *dev_flipped, hipStream_t stream);
extern "C" int vg_batch_i8_launch(const uint8_t *dev_rows_signed, long long n_rows, long long stride_bytes,
                                  const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root, int is_u8,
                                  const uint32_t *dev_row_stat, uint64_t *dev_cand, int npart,
                                  int tiles_per_part, uint64_t *dev_out_keys, hipStream_t stream);

// ---- f16 / bf16 batches: matrix-core filter + exact f64 re-evaluation (vg_batch_h.hip)
extern "C" size_t vg_batch_h_lds_bytes(long long stride_bytes, int k);
extern "C" int vg_batch_h_queries_per_block(long long stride_bytes);
extern "C" int vg_batch_h_plan(long long stride_bytes, int k, int nq, int *waves, int *blocks_per_cu);
extern "C" int vg_batch_h_launch(const uint8_t *dev_rows, int rows_tiled, long long n_rows, long long stride_bytes, int dim, int type_code,
                                 const uint8_t *dev_xrows, long long xstride_bytes,
                                 const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root,
                                 const float *dev_row_nn, uint64_t *dev_cand, int npart, int tiles_per_part,
                                 uint64_t *dev_out_keys, unsigned long long *dev_evals, int waves,
                                 uint64_t *dev_pairs, uint32_t *dev_pair_counts, int pair_cap, hipStream_t stream);
extern "C" int vg_batch_h_split(long long stride_bytes, int k);        // vg_batch_h.hip: the split form is on for such rows
#define VG_BPAIR_CAP 2048               // candidate pairs per filter wavefront and stage (a batch that overflows one falls back to the fused kernel)
#define VG_BPAIR_CAP_LONG 8192          // ... of the long-row kernel (64 KB per region, 1024 regions: its fallback is one scan per query)
extern "C" int vg_tile_major_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, uint8_t *dev_out, hipStream_t stream);
extern "C" int vg_f32_to_bf16_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int dim,
                                     uint8_t *dev_out, long long ostride, hipStream_t stream);

// f32 corpora with rows of 513 .. 1024 floats have no f32 matrix-core kernel (A does not fit the register file, the tiles not
// the LDS): they run through the half-precision kernel with a bf16 SHADOW copy as the filter's input and the f32 rows for the
// exact evaluation.  Shorter rows go the same way (5x the f32 MFMA kernel at D = 384: the filter runs at the bf16 rate over
// half the bytes, the survivors carry the single-query kernel's f32 arithmetic) when the corpus has - or by the filter scan's
// own rule (vg_filter.hip: switched on, >= 3 GB) is about to have - that shadow copy; VG_F32_FILTER=1 / 0 forces it on / off.
// A selectivity guard like the filter scan's watches the exact evaluations (see scan_topk_batch_mfma).
long long vg_bf16_shadow_stride(const vg_corpus *c) { return (((long long)c->dim * 2 + 15) / 16) * 16; }
static long long bf16_shadow_stride(const vg_corpus *c) { return vg_bf16_shadow_stride(c); }
static bool batch_f32_filter_short_rows(const vg_corpus *c) {
    const int sw = vg_sw(SW_VG_F32_FILTER, -1);
    if (sw >= 0) return sw != 0;
    return vg_scan_filter_policy(c) && c->bfilter_cooldown == 0;
}
static bool batch_f32_filter_eligible(const vg_corpus *c, int metric, int k) {
    if (vg_sw(SW_VG_BATCH_MFMA, 1) == 0 || c->vtype != VG_TYPE_F32 || metric == VG_DIST_L1) return false;
    if (c->dim <= 512 && !batch_f32_filter_short_rows(c)) return false;
    return vg_batch_h_lds_bytes(bf16_shadow_stride(c), k) != 0;
}
int vg_ensure_bf16_shadow(vg_corpus *c) {
    const long long bs = bf16_shadow_stride(c);
    if (c->bf_cap < c->n_rows) {
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rows_bf) hipFree(c->d_rows_bf);
        c->d_rows_bf = nullptr; c->bf_cap = 0; c->bf_rows = 0;
        HIP_TRY(hipMalloc(&c->d_rows_bf, (size_t)cap * bs));
        c->bf_cap = cap;
    }
    if (c->bf_rows < c->n_rows) {
        int rc = vg_f32_to_bf16_launch(c->d_rows, c->bf_rows, c->n_rows - c->bf_rows, c->stride, c->dim, c->d_rows_bf, bs, c->stream);
        if (rc != 0) return vg_fail(VG_ERR_HIP, "bf16 shadow pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->bf_rows = c->n_rows;
    }
    return VG_OK;
}

// f32 corpora through the bf16 filter: the shadow copy the BATCH kernel streams is tile-major (vg_f32_to_bf16_tm_kernel) - the
// row-major bf16 copy above is what the single-query bf16 filter scan reads (VG_SCAN_FILTER_SHADOW=bf16; the default single-query
// filter streams the int8 copy, so by default only this one exists).  Same + 50 % of the corpus; tm_rows / tm_cap are shared with
// the f16 / bf16 corpora's tile-major copy (a corpus has one element type).  -1: switched off / no memory (row-major gather).
extern "C" int vg_f32_to_bf16_tm_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int dim,
                                        uint8_t *dev_out, long long ostride, hipStream_t stream);
static int ensure_bf16_tile_major(vg_corpus *c) {
    if (vg_sw(SW_VG_BATCH_TILE_MAJOR, 1) == 0 || c->tm_disabled) return -1;
    const long long bs = bf16_shadow_stride(c);
    if (c->tm_cap < c->n_rows) {
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rows_tm) hipFree(c->d_rows_tm);
        c->d_rows_tm = nullptr; c->tm_cap = 0; c->tm_rows = 0;
        const size_t bytes = (size_t)((cap + 31) / 32 * 32) * bs;                               // whole tiles
        if (hipMalloc(&c->d_rows_tm, bytes) != hipSuccess) { (void)hipGetLastError(); c->d_rows_tm = nullptr; c->tm_disabled = true; return -1; }
        HIP_TRY(hipMemsetAsync(c->d_rows_tm, 0, bytes, c->stream));                             // (rows past the end: defined bytes)
        c->tm_cap = cap;
    }
    if (c->tm_rows < c->n_rows) {
        int rc = vg_f32_to_bf16_tm_launch(c->d_rows, c->tm_rows, c->n_rows - c->tm_rows, c->stride, c->dim, c->d_rows_tm, bs, c->stream);
        if (rc != 0) return vg_fail(VG_ERR_HIP, "tile-major bf16 shadow pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->tm_rows = c->n_rows;
    }
    return VG_OK;
}

// what the batched bf16 filter streams for an f32 corpus: the tile-major shadow copy, else (switched off / no room) the row-major one
static int ensure_f32_batch_shadow(vg_corpus *c, const uint8_t **rows, int *tiled) {
    const int rct = ensure_bf16_tile_major(c);
    if (rct == VG_OK) { *rows = c->d_rows_tm; *tiled = 1; return VG_OK; }
    if (rct != -1) return rct;
    const int rc = vg_ensure_bf16_shadow(c);
    if (rc != VG_OK) return rc;
    *rows = c->d_rows_bf; *tiled = 0;
    return VG_OK;
}

// f16 / bf16 corpora: the tile-major copy the matrix-core kernel streams (+ 100 % of the corpus in HBM, made at the first batch and
// extended per appended row; without it every LDS-DMA instruction gathers 32-byte runs from 32 rows - vg_batch_i8.hip).  A corpus
// it does not fit next to keeps the row-major gather.
static int ensure_half_tile_major(vg_corpus *c) {
    if (vg_sw(SW_VG_BATCH_TILE_MAJOR, 1) == 0 || c->tm_disabled) return -1;
    if (c->tm_cap < c->n_rows) {
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rows_tm) hipFree(c->d_rows_tm);
        c->d_rows_tm = nullptr; c->tm_cap = 0; c->tm_rows = 0;
        const size_t bytes = (size_t)((cap + 31) / 32 * 32) * c->stride;                        // whole tiles
        if (hipMalloc(&c->d_rows_tm, bytes) != hipSuccess) { (void)hipGetLastError(); c->d_rows_tm = nullptr; c->tm_disabled = true; return -1; }
        HIP_TRY(hipMemsetAsync(c->d_rows_tm, 0, bytes, c->stream));                             // (rows past the end: defined bytes)
        c->tm_cap = cap;
    }
    if (c->tm_rows < c->n_rows) {
        int rc = vg_tile_major_launch(c->d_rows, c->tm_rows, c->n_rows - c->tm_rows, c->stride, c->d_rows_tm, c->stream);
        if (rc != 0) return vg_fail(VG_ERR_HIP, "tile-major pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->tm_rows = c->n_rows;
    }
    return VG_OK;
}

static bool batch_h_eligible(const vg_corpus *c, int metric, int k) {
    if (vg_sw(SW_VG_BATCH_MFMA, 1) == 0) return false;
    if (c->vtype != VG_TYPE_F16 && c->vtype != VG_TYPE_BF16) return false;
    if (metric == VG_DIST_L1) return false;
    return vg_batch_h_lds_bytes(c->stride, k) != 0;
}

static bool batch_i8_eligible(const vg_corpus *c, int metric, int k) {
    if (vg_sw(SW_VG_BATCH_MFMA, 1) == 0) return false;
    if (c->vtype != VG_TYPE_U8 && c->vtype != VG_TYPE_I8) return false;
    if (metric == VG_DIST_L1) return false;
    return vg_batch_i8_lds_bytes(c->stride, k) != 0;
}

// row sums + the TILE-MAJOR copy of the corpus the matrix-core kernel streams (uint8: XOR 0x80 = the signed representation the
// instruction multiplies), once per appended row.  Tile t = rows 32t .. 32t+31 occupies 32 * stride contiguous bytes laid out
// exactly like the kernel's LDS tile (chunk column c of the 32 rows = 512 contiguous bytes at c * 512 + row * 16), so that every
// LDS-DMA instruction of the kernel moves 1 KiB of CONTIGUOUS memory.  Gathering 32-byte runs from 32 rows of the row-major
// corpus instead held the DMA path at ~13 GB/s per CU - what bounded the kernel (vg_batch_i8.hip).
static int ensure_i8_row_stats(vg_corpus *c) {
    const bool u8 = (c->vtype == VG_TYPE_U8);
    if (c->i8_cap < c->n_rows) {
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_sx) hipFree(c->d_sx);
        if (c->d_rows_s8) hipFree(c->d_rows_s8);
        c->d_sx = nullptr; c->d_rows_s8 = nullptr; c->i8_cap = 0; c->i8_rows = 0;
        // (sum x, sum x^2) pairs; + slack: the batch kernel fetches the sums of four whole 32-row tiles at a time, also behind the last row
        HIP_TRY(hipMalloc(&c->d_sx, (size_t)(cap + 256) * 2 * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->d_sx, 0, (size_t)(cap + 256) * 2 * sizeof(uint32_t), c->stream));
        const size_t tiled_bytes = (size_t)((cap + 31) / 32 * 32) * c->stride;               // whole tiles
        HIP_TRY(hipMalloc(&c->d_rows_s8, tiled_bytes));
        HIP_TRY(hipMemsetAsync(c->d_rows_s8, 0, tiled_bytes, c->stream));                  // (rows past the end: defined bytes)
        c->i8_cap = cap;
    }
    if (c->i8_rows < c->n_rows) {
        int rc = vg_i8_rowstat_launch(c->d_rows, c->i8_rows, c->n_rows - c->i8_rows, c->stride, u8 ? 1 : 0, c->d_sx,
                                      c->d_rows_s8, c->stream);
        if (rc != 0) return vg_fail(VG_ERR_HIP, "row-statistics pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->i8_rows = c->n_rows;
    }
    return VG_OK;
}

// ---- rows of 1025 .. 3072 elements (f16 / bf16 corpora; f32 corpora through their bf16 shadow copy): vg_batch_hl.hip, the K dimension
// split over the four wavefronts of a workgroup.  Always the split form (filter kernel -> candidate pairs -> exact evaluation) and always
// from the tile-major copy; a batch whose pairs overflow a region (data the filter cannot separate, queries with Inf / NaN) is answered by
// the multi-query scan, and so are the next ones over this corpus.
extern "C" int vg_batch_hl_serves(long long stride_bytes, int k);
extern "C" int vg_batch_hl_queries_per_block(long long stride_bytes);
extern "C" int vg_batch_hl_regions(long long stride_bytes, int nq_pad, int npart);
extern "C" int vg_batch_hl_launch(const uint8_t *dev_rows, long long n_rows, long long stride_bytes, int dim, int type_code,
                                  const uint8_t *dev_xrows, long long xstride_bytes,
                                  const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root,
                                  const float *dev_row_nn, const float *dev_query_nn, uint64_t *dev_cand, int npart,
                                  uint64_t *dev_out_keys, unsigned long long *dev_evals,
                                  uint64_t *dev_pairs, uint32_t *dev_pair_counts, int pair_cap, hipStream_t stream);
extern "C" int vg_batch_hl_query_norms(const uint8_t *dev_queries, long long stride_bytes, int dim, int type_code, int nq_real, int nq_pad,
                                       float *dev_out, hipStream_t stream);
static long long batch_long_stride(const vg_corpus *c) { return c->vtype == VG_TYPE_F32 ? bf16_shadow_stride(c) : c->stride; }
static bool batch_long_eligible(const vg_corpus *c, int metric, int k) {
    if (vg_sw(SW_VG_BATCH_MFMA, 1) == 0 || vg_sw(SW_VG_BATCH_LONG, 1) == 0 || metric == VG_DIST_L1 || c->tm_disabled) return false;
    if (c->vtype != VG_TYPE_F32 && c->vtype != VG_TYPE_F16 && c->vtype != VG_TYPE_BF16) return false;
    return vg_batch_hl_serves(batch_long_stride(c), k) != 0;
}

static int scan_topk_batch_long(vg_corpus *c, int metric, const void *queries, int nq, int k, uint64_t *out_keys, int *out_counts) {
    const bool f32 = (c->vtype == VG_TYPE_F32);
    const long long fstride = batch_long_stride(c);
    const int QPB = vg_batch_hl_queries_per_block(fstride);
    const int nq_pad = ((nq + QPB - 1) / QPB) * QPB;
    const int G = nq_pad / QPB;
    // partitions: one workgroup per CU and query group (VG_BATCH_LONG_BPC rounds of them: measured, 1024 x 2M x 1536 - 1: 11.0 ms, 2: 11.9,
    // 4: 14.7, 8: 20.9 - a workgroup's set-up, the A operand of 64 queries, is paid per partition)
    int npart = std::max(1, c->cu_count * std::max(1, vg_sw(SW_VG_BATCH_LONG_BPC, 1)) / G);
    if (npart >= 8) npart = (npart / 8) * 8;
    npart = std::min(npart, 256);
    const long long ntiles = (c->n_rows + 31) / 32;
    npart = (int)std::min<long long>(npart, ntiles);
    int rcn = vg_ensure_row_norms(c);
    if (rcn != VG_OK) return rcn;
    rcn = f32 ? ensure_bf16_tile_major(c) : ensure_half_tile_major(c);
    if (rcn == -1) return -1;                              // (no room for the copy: the multi-query scan)
    if (rcn != VG_OK) return rcn;
    const size_t qrows = (size_t)nq_pad * c->stride, qbytes = qrows + (size_t)nq_pad * sizeof(float);      // the query rows, then their norms
    const size_t candbytes = (size_t)nq_pad * vg_batch_lists_per_query(c->n_rows, npart) * 64 * sizeof(uint64_t);
    const size_t keybytes = (size_t)nq_pad * 64 * sizeof(uint64_t);
    if (c->bq_bytes < qbytes) { if (c->d_bq) hipFree(c->d_bq); c->d_bq = nullptr; c->bq_bytes = 0;
                                HIP_TRY(hipMalloc(&c->d_bq, qbytes)); c->bq_bytes = qbytes; }
    if (c->bcand_bytes < candbytes) { if (c->d_bcand) hipFree(c->d_bcand); c->d_bcand = nullptr; c->bcand_bytes = 0;
                                      HIP_TRY(hipMalloc(&c->d_bcand, candbytes)); c->bcand_bytes = candbytes; }
    if (c->bkeys_bytes < keybytes) { if (c->d_bkeys) hipFree(c->d_bkeys); c->d_bkeys = nullptr; c->bkeys_bytes = 0;
                                     HIP_TRY(hipMalloc(&c->d_bkeys, keybytes)); c->bkeys_bytes = keybytes; }
    const int n_regions = vg_batch_hl_regions(fstride, nq_pad, npart);
    const size_t need = (size_t)n_regions * VG_BPAIR_CAP_LONG * sizeof(uint64_t), needc = ((size_t)n_regions + 1) * sizeof(uint32_t);
    if (c->bpairs_bytes < need) { if (c->d_bpairs) hipFree(c->d_bpairs); c->d_bpairs = nullptr; c->bpairs_bytes = 0;
                                  HIP_TRY(hipMalloc(&c->d_bpairs, need)); c->bpairs_bytes = need; }
    if (c->bpcount_bytes < needc) { if (c->d_bpcounts) hipFree(c->d_bpcounts); c->d_bpcounts = nullptr; c->bpcount_bytes = 0;
                                    HIP_TRY(hipMalloc(&c->d_bpcounts, needc)); c->bpcount_bytes = needc; }
    // the queries go up through a PINNED buffer of the corpus (zero-padded rows of the corpus stride, zero rows up to nq_pad; the
    // per-query norms come back through its tail): a pageable source is staged by the runtime at a fraction of the link's rate, and
    // at 1024 x 6 KB that, a zero-filled vector and a scalar norm loop on the host were 1.35 ms of every batch (profiles/r8l)
    const size_t row_bytes = (size_t)c->dim * c->es;
    if (c->h_bq_bytes < qbytes) {
        if (c->h_bq) hipHostFree(c->h_bq);
        c->h_bq = nullptr; c->h_bq_bytes = 0;
        HIP_TRY(hipHostMalloc(&c->h_bq, qbytes));
        c->h_bq_bytes = qbytes;
    }
    if (row_bytes == (size_t)c->stride) memcpy(c->h_bq, queries, (size_t)nq * row_bytes);
    else
        for (int i = 0; i < nq; ++i) {
            memcpy(c->h_bq + (size_t)i * c->stride, (const uint8_t *)queries + (size_t)i * row_bytes, row_bytes);
            memset(c->h_bq + (size_t)i * c->stride + row_bytes, 0, (size_t)c->stride - row_bytes);
        }
    if (nq_pad > nq) memset(c->h_bq + (size_t)nq * c->stride, 0, (size_t)(nq_pad - nq) * c->stride);
    HIP_TRY(hipMemcpyAsync(c->d_bq, c->h_bq, qrows, hipMemcpyHostToDevice, c->stream));
    // sum q^2 per query, on the device.  A query the filter cannot judge (Inf / NaN elements, a norm of zero or out of range) would
    // send EVERY row down the exact path - in the split form that is a pair per row: its norm comes out as -1, the kernels multiply it as
    // zero and let none of its pairs pass, and it is answered by a single scan below
    float *dev_qnn = reinterpret_cast<float *>((uint8_t *)c->d_bq + qrows);
    {
        const int rcq = vg_batch_hl_query_norms((const uint8_t *)c->d_bq, c->stride, c->dim, f32 ? 2 : (c->vtype == VG_TYPE_BF16 ? 1 : 0), nq, nq_pad, dev_qnn, c->stream);
        if (rcq != 0) return vg_fail(VG_ERR_HIP, "query norm pass failed: %s", hipGetErrorString((hipError_t)rcq));
    }
    hipEvent_t *evs = nullptr;
    if (c->profiling) {
        int slot = (int)(c->prof_launches % VG_PROF_RING);
        evs = &c->ev[(size_t)slot * VG_PROF_EVS];
        c->ev_flags[(size_t)slot] = 0;
        ++c->prof_launches;
        hipEventRecord(evs[0], c->stream);
    }
    const int mode = metric == VG_DIST_DOT ? 0 : (metric == VG_DIST_COSINE ? 1 : 2), root = metric == VG_DIST_L2 ? 1 : 0;
    const int rc = vg_batch_hl_launch(c->d_rows_tm, c->n_rows, fstride, c->dim, f32 ? 2 : (c->vtype == VG_TYPE_BF16 ? 1 : 0), c->d_rows, c->stride,
                                      (const uint8_t *)c->d_bq, nq_pad, nq, k, mode, root, c->d_xnorm, dev_qnn,
                                      c->d_bcand, npart, c->d_bkeys, nullptr, c->d_bpairs, c->d_bpcounts, VG_BPAIR_CAP_LONG, c->stream);
    if (evs) { hipEventRecord(evs[2], c->stream); hipEventRecord(evs[3], c->stream); }
    if (rc == -1) { hipStreamSynchronize(c->stream); return -1; }
    if (rc != 0) return vg_fail(VG_ERR_HIP, "batched scan launch (long rows) failed: %s", hipGetErrorString((hipError_t)rc));
    uint32_t overflow = 0;
    std::vector<uint64_t> keys((size_t)nq * 64);
    HIP_TRY(hipMemcpyAsync(&overflow, c->d_bpcounts + n_regions, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(keys.data(), c->d_bkeys, (size_t)nq * 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    float *h_qnn = reinterpret_cast<float *>(c->h_bq + qrows);
    HIP_TRY(hipMemcpyAsync(h_qnn, dev_qnn, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    vg_collect_timing(c);
    if (overflow != 0) { c->blong_cooldown = 16; return -1; }   // (rows the filter cannot separate / judge: the next batches scan)
    std::vector<int> unjudged;
    for (int i = 0; i < nq; ++i) if (h_qnn[i] < 0.0f) unjudged.push_back(i);
    for (int i = 0; i < nq; ++i) {
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            const uint64_t key = keys[(size_t)i * 64 + j];
            if (key == VG_EMPTY_KEY) break;
            out_keys[(size_t)i * k + cnt] = key;
            ++cnt;
        }
        out_counts[i] = cnt;
    }
    for (int i : unjudged) {
        const int rc1 = vg_scan_topk_keys(c, metric, (const uint8_t *)queries + (size_t)i * row_bytes, k, out_keys + (size_t)i * k, out_counts + i);
        if (rc1 != VG_OK) return rc1;
    }
    return VG_OK;
}

// ---- f32 corpora, large batches: the INTEGER matrix cores as the filter over the int8 shadow copy (vg_batch_q8.hip) - 64 queries per
// wavefront, a quarter of the corpus' bytes streamed, the single scan's f32 arithmetic for the pairs that pass.  Default for every batch
// (round 5: from 257 queries on) over corpora the filter scans' policy covers; VG_BATCH_Q8=0 / 1 forces it off / on.
extern "C" int vg_batch_q8_serves(long long q8stride_bytes, long long xstride_bytes, int k);
extern "C" int vg_batch_q8_queries_per_block(void);
extern "C" int vg_batch_q8_padded_queries(int nq, long long q8stride_bytes);
extern "C" int vg_batch_q8_max_partitions(void);
extern "C" int vg_batch_q8_regions(int nq_pad, int npart);
extern "C" size_t vg_batch_q8_work_bytes(int nq_pad, long long q8stride_bytes, long long xstride_bytes);
extern "C" size_t vg_batch_q8_work_stat_offset(int nq_pad, long long q8stride_bytes, long long xstride_bytes);
extern "C" size_t vg_batch_q8_work_perm_offset(int nq_pad, long long q8stride_bytes, long long xstride_bytes);
extern "C" int vg_batch_q8_max_queries(void);
extern "C" int vg_q8_rstat_launch(const void *dev_q8stat, const float *dev_xnorm, long long row0, long long n, void *dev_out, int nn_squared, hipStream_t stream);
extern "C" int vg_batch_q8_launch(const uint8_t *dev_rows_tm, const void *dev_rstat, long long n_rows, long long q8stride, int dim,
                                  const uint8_t *dev_xrows, long long xstride, const float *dev_xnorm,
                                  const uint8_t *dev_xqueries, void *dev_qwork, int nq_pad, int nq_real, int k, int mode, int root,
                                  uint64_t *dev_cand, int npart, uint64_t *dev_out_keys, unsigned long long *dev_evals,
                                  uint64_t *dev_pairs, uint32_t *dev_pair_counts, int pair_cap, int type_code, hipStream_t stream);
extern "C" int vg_batch_q8_workgroups_per_cu(long long q8stride_bytes);
extern "C" size_t vg_batch_q8_pack_bytes(int nq_pad, int k);
extern "C" int vg_batch_q8_pack_launch(const uint64_t *dev_keys, const void *dev_qwork, int nq_pad, long long q8stride, long long xstride, int k,
                                       const uint32_t *dev_overflow_flag, const unsigned long long *dev_evals, void *dev_out, hipStream_t stream);
static long long q8_shadow_stride_of(const vg_corpus *c) { return (((long long)c->dim + 15) / 16) * 16; }
static bool batch_q8_eligible(const vg_corpus *c, int metric, int k, int nq) {
    const bool served_type = c->vtype == VG_TYPE_F32 || c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16;   // (the int8 image of the row, whatever it is stored as)
    if (vg_sw(SW_VG_BATCH_MFMA, 1) == 0 || !served_type || metric == VG_DIST_L1 || c->q8tm_disabled || c->filter_disabled) return false;
    const int sw = vg_sw(SW_VG_BATCH_Q8, -1);
    if (sw != 1 && c->vtype == VG_TYPE_F32 && vg_sw(SW_VG_F32_FILTER, -1) == 0) return false;    // VG_F32_FILTER=0: f32 batches unfiltered (the f32 matrix-core kernel)
    if (sw == 0 || c->n_rows < (sw == 1 ? (1ll << 16) : (1ll << 20))) return false;     // (forced: from 2048 tiles on - the tests' sizes)
    // (round 6: every batch size - with the all-padding-set fix a 4-query batch takes 1.3 ms here against 2.2 through the bf16 filter, 128
    //  queries 1.6 against 2.4, 256 queries 2.0 against 3.0; profiles/r10_small_batches_int8_vs_bf16_filter.txt)
    if (sw < 0 && !vg_scan_filter_policy(c)) return false;             // (its own overflow guard: bq8_cooldown; the bf16 filter's does not apply)
    return vg_batch_q8_serves(q8_shadow_stride_of(c), c->stride, k) != 0;
}
// the tile-major int8 copy + its per-row statistics, built from the row-major shadow copy (vg_filter.hip) and the cached norms, extended
// per appended row; -1: no room (the caller takes another path)
static int ensure_q8_tile_major(vg_corpus *c) {
    int rc = vg_ensure_row_norms(c);
    if (rc == VG_OK) rc = vg_ensure_q8_shadow(c);
    if (rc == VG_ERR_NOMEM) { (void)hipGetLastError(); c->q8tm_disabled = true; return -1; }
    if (rc != VG_OK) return rc;
    const long long qs = q8_shadow_stride_of(c);
    if (c->q8tm_cap < c->n_rows) {
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rows_q8tm) hipFree(c->d_rows_q8tm);
        if (c->d_q8tm_stat) hipFree(c->d_q8tm_stat);
        c->d_rows_q8tm = nullptr; c->d_q8tm_stat = nullptr; c->q8tm_cap = 0; c->q8tm_rows = 0;
        const size_t tiles = (size_t)((cap + 31) / 32);
        const size_t bytes = tiles * 32 * (size_t)qs, sbytes = (tiles + 2) * 32 * 16;
        if (hipMalloc(&c->d_rows_q8tm, bytes) != hipSuccess || hipMalloc(&c->d_q8tm_stat, sbytes) != hipSuccess) {
            (void)hipGetLastError();
            if (c->d_rows_q8tm) hipFree(c->d_rows_q8tm);
            c->d_rows_q8tm = nullptr; c->d_q8tm_stat = nullptr; c->q8tm_disabled = true;
            return -1;
        }
        HIP_TRY(hipMemsetAsync(c->d_rows_q8tm, 0, bytes, c->stream));                          // (rows past the end: defined bytes)
        HIP_TRY(hipMemsetAsync(c->d_q8tm_stat, 0, sbytes, c->stream));
        c->q8tm_cap = cap;
    }
    if (c->q8tm_rows < c->n_rows) {
        const long long n = c->n_rows - c->q8tm_rows;
        rc = vg_tile_major_launch(c->d_rows_q8, c->q8tm_rows, n, qs, c->d_rows_q8tm, c->stream);
        if (rc == 0) rc = vg_q8_rstat_launch(c->d_q8stat, c->d_xnorm, c->q8tm_rows, n, c->d_q8tm_stat, c->vtype == VG_TYPE_F32 ? 0 : 1, c->stream);
        if (rc != 0) return vg_fail(VG_ERR_HIP, "tile-major int8 pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->q8tm_rows = c->n_rows;
    }
    return VG_OK;
}

static int scan_topk_batch_q8(vg_corpus *c, int metric, const void *queries, int nq, int k, uint64_t *out_keys, int *out_counts) {
    const long long qs = q8_shadow_stride_of(c);
    const int QPB = vg_batch_q8_queries_per_block();
    const int nq_pad = vg_batch_q8_padded_queries(nq, qs);                  // (128 slots for up to 128 queries over short rows, whole 256-query workgroups otherwise)
    const int G = std::max(1, nq_pad / QPB);
    // two 4-wavefront workgroups per CU (long rows: one of 8); the 128-slot form has ONE query group: 512 partitions fill both workgroup slots of
    // every CU (the stages' merges see np / 4 lists per query)
    const int npart = std::min((nq_pad <= 256 && vg_batch_q8_workgroups_per_cu(qs) == 2) ? vg_batch_q8_max_partitions() : 256, std::max(8, (vg_batch_q8_workgroups_per_cu(qs) * c->cu_count / G) / 8 * 8));
    int rcn = ensure_q8_tile_major(c);
    if (rcn == -1) { c->bq8_status = 1; return -1; }
    if (rcn != VG_OK) return rcn;
    rcn = vg_ensure_filter_counters(c);
    if (rcn != VG_OK) return rcn;
    const size_t qrows = (size_t)nq_pad * c->stride, qbytes = qrows + vg_batch_q8_work_bytes(nq_pad, qs, c->stride);
    const size_t candbytes = (size_t)nq_pad * std::max(npart, 64) * 64 * sizeof(uint64_t);
    const size_t keybytes = (size_t)nq_pad * 64 * sizeof(uint64_t);
    if (c->bq_bytes < qbytes) { if (c->d_bq) hipFree(c->d_bq); c->d_bq = nullptr; c->bq_bytes = 0;
                                HIP_TRY(hipMalloc(&c->d_bq, qbytes)); c->bq_bytes = qbytes; }
    if (c->bcand_bytes < candbytes) { if (c->d_bcand) hipFree(c->d_bcand); c->d_bcand = nullptr; c->bcand_bytes = 0;
                                      HIP_TRY(hipMalloc(&c->d_bcand, candbytes)); c->bcand_bytes = candbytes; }
    if (c->bkeys_bytes < keybytes) { if (c->d_bkeys) hipFree(c->d_bkeys); c->d_bkeys = nullptr; c->bkeys_bytes = 0;
                                     HIP_TRY(hipMalloc(&c->d_bkeys, keybytes)); c->bkeys_bytes = keybytes; }
    const int n_regions = vg_batch_q8_regions(nq_pad, npart);
    const size_t need = (size_t)n_regions * VG_BPAIR_CAP * sizeof(uint64_t), needc = ((size_t)n_regions + 1) * sizeof(uint32_t);
    if (c->bpairs_bytes < need) { if (c->d_bpairs) hipFree(c->d_bpairs); c->d_bpairs = nullptr; c->bpairs_bytes = 0;
                                  HIP_TRY(hipMalloc(&c->d_bpairs, need)); c->bpairs_bytes = need; }
    if (c->bpcount_bytes < needc) { if (c->d_bpcounts) hipFree(c->d_bpcounts); c->d_bpcounts = nullptr; c->bpcount_bytes = 0;
                                    HIP_TRY(hipMalloc(&c->d_bpcounts, needc)); c->bpcount_bytes = needc; }
    // the f32 queries go up through the corpus' pinned buffer: zero-padded rows of the corpus stride, zero rows up to nq_pad; the queries'
    // statistics (which of them the filter could judge) come back through its tail
    const size_t packbytes = vg_batch_q8_pack_bytes(nq_pad, k);                              // what comes back, packed by the device: one copy
    const size_t row_bytes = (size_t)c->dim * c->es, statbytes = (packbytes + 15) / 16 * 16;
    if (c->h_bq_bytes < qrows + statbytes) {
        if (c->h_bq) hipHostFree(c->h_bq);
        c->h_bq = nullptr; c->h_bq_bytes = 0;
        HIP_TRY(hipHostMalloc(&c->h_bq, qrows + statbytes));
        c->h_bq_bytes = qrows + statbytes;
    }
    if (row_bytes == (size_t)c->stride) memcpy(c->h_bq, queries, (size_t)nq * row_bytes);
    else
        for (int i = 0; i < nq; ++i) {
            memcpy(c->h_bq + (size_t)i * c->stride, (const uint8_t *)queries + (size_t)i * row_bytes, row_bytes);
            memset(c->h_bq + (size_t)i * c->stride + row_bytes, 0, (size_t)c->stride - row_bytes);
        }
    if (nq_pad > nq) memset(c->h_bq + (size_t)nq * c->stride, 0, (size_t)(nq_pad - nq) * c->stride);
    HIP_TRY(hipMemcpyAsync(c->d_bq, c->h_bq, qrows, hipMemcpyHostToDevice, c->stream));
    hipEvent_t *evs = nullptr;
    if (c->profiling) {
        int slot = (int)(c->prof_launches % VG_PROF_RING);
        evs = &c->ev[(size_t)slot * VG_PROF_EVS];
        c->ev_flags[(size_t)slot] = 0;
        ++c->prof_launches;
        hipEventRecord(evs[0], c->stream);
    }
    const int mode = metric == VG_DIST_DOT ? 0 : (metric == VG_DIST_COSINE ? 1 : 2), root = metric == VG_DIST_L2 ? 1 : 0;
    uint8_t *qwork = (uint8_t *)c->d_bq + qrows;
    const int rc = vg_batch_q8_launch(c->d_rows_q8tm, c->d_q8tm_stat, c->n_rows, qs, c->dim, c->d_rows, c->stride, c->d_xnorm,
                                      (const uint8_t *)c->d_bq, qwork, nq_pad, nq, k, mode, root, c->d_bcand, npart, c->d_bkeys,
                                      c->d_filter_evals + 1, c->d_bpairs, c->d_bpcounts, VG_BPAIR_CAP,
                                      c->vtype == VG_TYPE_F32 ? 2 : (c->vtype == VG_TYPE_BF16 ? 1 : 0), c->stream);
    if (evs) { hipEventRecord(evs[2], c->stream); hipEventRecord(evs[3], c->stream); }
    if (rc == -1) { hipStreamSynchronize(c->stream); c->bq8_status = 2; return -1; }
    if (rc != 0) return vg_fail(VG_ERR_HIP, "batched scan launch (int8 filter) failed: %s", hipGetErrorString((hipError_t)rc));
    // the lists come back in SORTED SLOT order (vg_batch_q8.hip sorts the batch by the int8 images' norms): slot p answers query perm[p];
    // overflow flag, evaluation counter, permutation, judged flags and the k keys of every slot packed by the device into one buffer
    // (straight into the pinned, device-mapped buffer - like the single scans' h_keys: a copy command behind the last kernel started ~0.17 ms late)
    uint32_t *h_pack = reinterpret_cast<uint32_t *>(c->h_bq + qrows);
    int rcp = vg_batch_q8_pack_launch(c->d_bkeys, qwork, nq_pad, qs, c->stride, k, c->d_bpcounts + n_regions, c->d_filter_evals + 1, h_pack, c->stream);
    if (rcp != 0) return vg_fail(VG_ERR_HIP, "batched scan (int8 filter): result pack failed: %s", hipGetErrorString((hipError_t)rcp));
    HIP_TRY(hipStreamSynchronize(c->stream));
    vg_collect_timing(c);
    c->h_filter_evals[1] = (unsigned long long)h_pack[2] | ((unsigned long long)h_pack[3] << 32);
    if (h_pack[0] != 0) { c->bq8_cooldown = 16; c->bq8_status = 3; return -1; }   // (data the bound cannot separate: the next batches take the other paths)
    c->bq8_status = 0;
    const int *h_perm = reinterpret_cast<const int *>(h_pack + 4);
    const uint32_t *h_judged = h_pack + 4 + nq_pad;
    const uint64_t *keys = reinterpret_cast<const uint64_t *>(h_pack + 4 + 2 * (size_t)nq_pad);
    std::vector<int> unjudged;
    for (int p = 0; p < nq_pad; ++p) {
        const int i = h_perm[p];
        if (i < 0 || i >= nq) continue;                                 // (padding)
        if (h_judged[p] == 0u) { unjudged.push_back(i); continue; }
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            const uint64_t key = keys[(size_t)p * k + j];
            if (key == VG_EMPTY_KEY) break;
            out_keys[(size_t)i * k + cnt] = key;
            ++cnt;
        }
        out_counts[i] = cnt;
    }
    for (int i : unjudged) {                                            // queries the filter could not judge (Inf / NaN / zero / out of range): a single scan each
        const int rc1 = vg_scan_topk_keys(c, metric, (const uint8_t *)queries + (size_t)i * row_bytes, k, out_keys + (size_t)i * k, out_counts + i);
        if (rc1 != VG_OK) return rc1;
    }
    return VG_OK;
}

static bool batch_mfma_eligible(const vg_corpus *c, int metric, int k) {
    if (vg_sw(SW_VG_BATCH_MFMA, 1) == 0) return false;
    if (c->vtype != VG_TYPE_F32) return false;
    if (metric == VG_DIST_L1) return false;                       // no matrix form
    return vg_batch_lds_bytes(c->stride, k) != 0;
}

static int scan_topk_batch_mfma(vg_corpus *c, int metric, const void *queries, int nq, int k, uint64_t *out_keys,
                                int *out_counts) {
    const bool quantized = (c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8);
    // f32 through the half-precision kernel (bf16 shadow copy): rows the f32 matrix-core kernel does not serve, or on request
    bool f32_filter = (c->vtype == VG_TYPE_F32) && batch_f32_filter_eligible(c, metric, k) &&
                      (vg_batch_lds_bytes(c->stride, k) == 0 || batch_f32_filter_short_rows(c));
    const bool f32_mfma_serves = (c->vtype == VG_TYPE_F32) && vg_batch_lds_bytes(c->stride, k) != 0;
    if (c->vtype == VG_TYPE_F32 && c->bfilter_cooldown > 0 && vg_sw(SW_VG_F32_FILTER, -1) < 0) --c->bfilter_cooldown;
    const uint8_t *f32_shadow = nullptr;                   // f32 through the bf16 filter: the shadow copy its matrix core reads
    int f32_shadow_tiled = 0;
    if (f32_filter && f32_mfma_serves) {
        // the shadow copy (+ 50 % of the corpus) and the norms must fit; a corpus they do not fit next to keeps the f32 kernel
        int rcs = vg_ensure_row_norms(c);
        if (rcs == VG_OK) rcs = ensure_f32_batch_shadow(c, &f32_shadow, &f32_shadow_tiled);
        if (rcs == VG_ERR_NOMEM) { (void)hipGetLastError(); c->filter_disabled = true; f32_filter = false; }
        else if (rcs != VG_OK) return rcs;
    }
    const bool half = (c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16) || f32_filter;
    c->last_batch_half = half;
    const long long fstride = f32_filter ? bf16_shadow_stride(c) : c->stride;       // row stride of what the matrix core reads
    // f16 / bf16 / f32-through-bf16 batches: the kernel comes as one 8-wavefront workgroup per CU or as two of four (vg_batch_h_plan)
    int h_waves = 8, h_bpc = 1;
    const bool h_split = half && vg_batch_h_split(fstride, k) != 0 && !c->bsplit_off;
    if (half && c->bsplit_off) {                           // (a batch overflowed the split form's pair buffer: the fused kernel, 8-wavefront form)
        h_waves = vg_batch_h_queries_per_block(fstride) / 32;
        if (vg_batch_h_lds_bytes(fstride, k) == 0) return -1;
    } else if (half && vg_batch_h_plan(fstride, k, nq, &h_waves, &h_bpc) != 0) return -1;
    uint32_t split_overflow = 0;
    bool split_used = false;
    const int QPB = quantized ? vg_batch_i8_queries_per_block(c->stride) : (half ? h_waves * 32 : 128);
    const int nq_pad = ((nq + QPB - 1) / QPB) * QPB;
    const int G = nq_pad / QPB;
    // partitions: enough workgroups to cover the chip (G * npart ~ CUs x workgroups per CU), a multiple of 8 (one per XCD), <= 256
    // (VG_BATCH_BPC overrides the workgroups per CU the partition count aims at)
    int npart = std::max(1, c->cu_count * std::max(1, vg_sw(SW_VG_BATCH_BPC, half ? h_bpc : 1)) / G);
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
    // queries: zero-padded rows of the corpus stride, zero rows up to nq_pad - through the corpus' pinned buffer (a pageable source is
    // staged by the runtime at a fraction of the link's rate)
    const size_t row_bytes = (size_t)c->dim * c->es;
    if (c->h_bq_bytes < qbytes) {
        if (c->h_bq) hipHostFree(c->h_bq);
        c->h_bq = nullptr; c->h_bq_bytes = 0;
        HIP_TRY(hipHostMalloc(&c->h_bq, qbytes));
        c->h_bq_bytes = qbytes;
    }
    if (row_bytes == (size_t)c->stride) memcpy(c->h_bq, queries, (size_t)nq * row_bytes);
    else
        for (int i = 0; i < nq; ++i) {
            memcpy(c->h_bq + (size_t)i * c->stride, (const uint8_t *)queries + (size_t)i * row_bytes, row_bytes);
            memset(c->h_bq + (size_t)i * c->stride + row_bytes, 0, (size_t)c->stride - row_bytes);
        }
    if (nq_pad > nq) memset(c->h_bq + (size_t)nq * c->stride, 0, (size_t)(nq_pad - nq) * c->stride);
    HIP_TRY(hipMemcpyAsync(c->d_bq, c->h_bq, qbytes, hipMemcpyHostToDevice, c->stream));
    if (quantized) {
        int rcn = ensure_i8_row_stats(c);
        if (rcn != VG_OK) return rcn;
    } else if (half || metric != VG_DIST_DOT) {            // f16 / bf16: every metric's filter needs sum x^2 per row
        int rcn = vg_ensure_row_norms(c);
        if (rcn != VG_OK) return rcn;
    }
    if (f32_filter && !f32_shadow) {
        const int rcn = ensure_f32_batch_shadow(c, &f32_shadow, &f32_shadow_tiled);
        if (rcn != VG_OK) return rcn;
    }

    hipEvent_t *evs = nullptr;
    if (c->profiling) {
        int slot = (int)(c->prof_launches % VG_PROF_RING);
        evs = &c->ev[(size_t)slot * VG_PROF_EVS];
        c->ev_flags[(size_t)slot] = 0;
        ++c->prof_launches;
        hipEventRecord(evs[0], c->stream);
    }
    const int mode = metric == VG_DIST_DOT ? 0 : (metric == VG_DIST_COSINE ? 1 : 2), root = metric == VG_DIST_L2 ? 1 : 0;
    const uint8_t *hrows = f32_filter ? f32_shadow : c->d_rows;         // what the half-precision kernel's matrix core reads
    int hrows_tiled = f32_filter ? f32_shadow_tiled : 0;
    if (half && !f32_filter) {
        const int rct = ensure_half_tile_major(c);
        if (rct == VG_OK) { hrows = c->d_rows_tm; hrows_tiled = 1; }
        else if (rct != -1) return rct;
    }
    int rc;
    if (quantized)
        rc = vg_batch_i8_launch(c->d_rows_s8, c->n_rows, c->stride, (const uint8_t *)c->d_bq,
                                nq_pad, nq, k, mode, root, c->vtype == VG_TYPE_U8 ? 1 : 0, c->d_sx, c->d_bcand, npart,
                                tiles_per_part, c->d_bkeys, c->stream);
    else if (half) {
        unsigned long long *dev_evals = nullptr;
        if (f32_filter) {                                  // the guard's counter (vg_filter.hip: slot [1] of the corpus' pair)
            const int rce = vg_ensure_filter_counters(c);
            if (rce != VG_OK) return rce;
            dev_evals = c->d_filter_evals + 1;
        }
        // the split form (vg_batch_h.hip, FILTER kind): one region of VG_BPAIR_CAP candidate pairs per filter wavefront + their counts
        uint64_t *dev_pairs = nullptr;
        uint32_t *dev_pair_counts = nullptr;
        const int n_regions = G * npart * h_waves;
        if (h_split && !c->bsplit_off) {
            const size_t need = (size_t)n_regions * VG_BPAIR_CAP * sizeof(uint64_t), needc = ((size_t)n_regions + 1) * sizeof(uint32_t);
            if (c->bpairs_bytes < need) { if (c->d_bpairs) hipFree(c->d_bpairs); c->d_bpairs = nullptr; c->bpairs_bytes = 0;
                                          if (hipMalloc(&c->d_bpairs, need) == hipSuccess) c->bpairs_bytes = need; else (void)hipGetLastError(); }
            if (c->bpcount_bytes < needc) { if (c->d_bpcounts) hipFree(c->d_bpcounts); c->d_bpcounts = nullptr; c->bpcount_bytes = 0;
                                            if (hipMalloc(&c->d_bpcounts, needc) == hipSuccess) c->bpcount_bytes = needc; else (void)hipGetLastError(); }
            if (c->bpairs_bytes >= need && c->bpcount_bytes >= needc) { dev_pairs = c->d_bpairs; dev_pair_counts = c->d_bpcounts; }
        }
        rc = vg_batch_h_launch(hrows, hrows_tiled, c->n_rows, fstride, c->dim,
                               f32_filter ? 2 : (c->vtype == VG_TYPE_BF16 ? 1 : 0), c->d_rows, c->stride, (const uint8_t *)c->d_bq,
                               nq_pad, nq, k, mode, root, c->d_xnorm, c->d_bcand, npart, tiles_per_part, c->d_bkeys, dev_evals, h_waves,
                               dev_pairs, dev_pair_counts, VG_BPAIR_CAP, c->stream);
        if (rc == 0 && dev_pairs) {                      // a region ran full (data the filter cannot separate): the fused kernel answers
            HIP_TRY(hipMemcpyAsync(&split_overflow, dev_pair_counts + n_regions, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            split_used = true;
        }
        if (rc == 0 && dev_evals)
            HIP_TRY(hipMemcpyAsync(c->h_filter_evals + 1, dev_evals, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    }
    else
        rc = vg_batch_launch((const float *)c->d_rows, c->n_rows, c->stride, (const float *)c->d_bq, nq_pad, nq, k, mode, root,
                             metric == VG_DIST_DOT ? nullptr : c->d_xnorm, c->d_bcand, npart, tiles_per_part, c->d_bkeys,
                             c->stream);
    if (evs) { hipEventRecord(evs[2], c->stream); hipEventRecord(evs[3], c->stream); }
    if (rc == -1) return -1;
    if (rc != 0) return vg_fail(VG_ERR_HIP, "batched scan launch failed: %s", hipGetErrorString((hipError_t)rc));
    std::vector<uint64_t> keys((size_t)nq * 64);
    HIP_TRY(hipMemcpyAsync(keys.data(), c->d_bkeys, (size_t)nq * 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    vg_collect_timing(c);
    if (split_used && split_overflow != 0) {               // the pair buffer ran full somewhere: this batch again through the fused kernel
        c->bsplit_off = true;                              // (and the next ones: such data keeps doing that)
        return scan_topk_batch_mfma(c, metric, queries, nq, k, out_keys, out_counts);
    }
    if (f32_filter && f32_mfma_serves) {
        // Selectivity guard.  An exact evaluation occupies a whole wavefront (and stalls its workgroup at the tile barrier):
        // the filter pays while few pairs need one.  On data it cannot separate (rows nearly identical to each other) nearly
        // every pair does; when a batch averaged more than one evaluation per 256 pairs the next 64 batches of this corpus
        // take the f32 matrix-core kernel, then the filter is tried again.  (Rows of 513+ floats have no such kernel: no guard.)
        const unsigned long long now = c->h_filter_evals[1];
        const unsigned long long pairs = (unsigned long long)nq * (unsigned long long)c->n_rows;
        if ((now - c->bfilter_evals_seen) > pairs / 256 && vg_sw(SW_VG_F32_FILTER, -1) < 0) c->bfilter_cooldown = 64;
        c->bfilter_evals_seen = now;
    }
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

// nq queries, NQ per pass of the multi-query scan kernel; all passes are enqueued back to back, one wait at the end
static int scan_topk_batch_multi(vg_corpus *c, int metric, const void *queries, int nq, int k, uint64_t *out_keys, int *out_counts) {
    const int NQ = vg_multi_queries_per_pass(c, metric);
    if (NQ == 0) return -1;
    const int ngroups = (nq + NQ - 1) / NQ, nq_pad = ngroups * NQ;
    const size_t qbytes = (size_t)nq_pad * c->stride, keybytes = (size_t)nq_pad * 64 * sizeof(uint64_t);
    if (c->bq_bytes < qbytes) { if (c->d_bq) hipFree(c->d_bq); c->d_bq = nullptr; c->bq_bytes = 0;
                                HIP_TRY(hipMalloc(&c->d_bq, qbytes)); c->bq_bytes = qbytes; }
    if (c->bkeys_bytes < keybytes) { if (c->d_bkeys) hipFree(c->d_bkeys); c->d_bkeys = nullptr; c->bkeys_bytes = 0;
                                     HIP_TRY(hipMalloc(&c->d_bkeys, keybytes)); c->bkeys_bytes = keybytes; }
    std::vector<uint8_t> hq(qbytes, 0);                       // zero-padded rows of the corpus stride; pad queries are zero
    const size_t row_bytes = (size_t)c->dim * c->es;
    for (int i = 0; i < nq; ++i) memcpy(hq.data() + (size_t)i * c->stride, (const uint8_t *)queries + (size_t)i * row_bytes, row_bytes);
    HIP_TRY(hipMemcpyAsync(c->d_bq, hq.data(), qbytes, hipMemcpyHostToDevice, c->stream));
    for (int g = 0; g < ngroups; ++g) {
        int rc = vg_launch_scan_multi(c, metric, (const uint8_t *)c->d_bq + (size_t)g * NQ * c->stride, k, c->d_cand,
                                      c->d_bkeys + (size_t)g * NQ * 64, c->stream);
        if (rc == -1) { hipStreamSynchronize(c->stream); return -1; }
        if (rc != VG_OK) return rc;
    }
    std::vector<uint64_t> keys((size_t)nq * 64);
    HIP_TRY(hipMemcpyAsync(keys.data(), c->d_bkeys, (size_t)nq * 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nq; ++i) {
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            const uint64_t key = keys[(size_t)i * 64 + j];
            if (key == VG_KEY_EMPTY) break;
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
    if (vg_metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_keys) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_batch_keys: NULL output");
    HIP_TRY(hipSetDevice(c->device));
    // A handful of queries over a corpus the filter scans serve: single scans (0.7 ms each at 10M x 384, whatever the type) beat one
    // 128- / 256-query-wide matrix pass (~2.9 ms however few of its query slots are used) up to three queries; they tie at four.
    const bool few = nq <= vg_sw(SW_VG_BATCH_MIN_QUERIES, 4) - 1 && vg_scan_filter_would_serve(c, metric, k);
    // (the int8 filter first: it serves rows up to 1536 elements too; what it hands back - or does not serve - goes on to the K-split bf16 kernel)
    if (!few && batch_q8_eligible(c, metric, k, nq) && c->bq8_cooldown > 0) --c->bq8_cooldown;
    else if (!few && batch_q8_eligible(c, metric, k, nq)) {
        const int slice = std::min(vg_batch_q8_max_queries(), std::max(512, vg_sw(SW_VG_BATCH_SLICE, 4096)));
        int rc = VG_OK;
        const size_t qbytes = (size_t)c->dim * c->es;
        for (int q0 = 0; q0 < nq && rc == VG_OK; q0 += slice) {
            const int nqs = std::min(slice, nq - q0);
            rc = scan_topk_batch_q8(c, metric, (const uint8_t *)queries + (size_t)q0 * qbytes, nqs, k, out_keys + (size_t)q0 * k, out_counts + q0);
        }
        if (rc != -1) { c->last_batch_path = 7; return rc; }
        for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    }
    if (!few && batch_long_eligible(c, metric, k) && c->blong_cooldown > 0) --c->blong_cooldown;
    else if (!few && batch_long_eligible(c, metric, k)) {
        const int slice = std::max(256, vg_sw(SW_VG_BATCH_SLICE, 4096));
        int rc = VG_OK;
        const size_t qbytes = (size_t)c->dim * c->es;
        for (int q0 = 0; q0 < nq && rc == VG_OK; q0 += slice) {
            const int nqs = std::min(slice, nq - q0);
            rc = scan_topk_batch_long(c, metric, (const uint8_t *)queries + (size_t)q0 * qbytes, nqs, k, out_keys + (size_t)q0 * k, out_counts + q0);
        }
        if (rc != -1) { c->last_batch_path = 4; return rc; }
        for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    }
    if (!few && (batch_mfma_eligible(c, metric, k) || batch_i8_eligible(c, metric, k) || batch_h_eligible(c, metric, k) ||
                 batch_f32_filter_eligible(c, metric, k))) {
        // very large batches go through in slices: the per-(query, partition) candidate lists are nq x ~128 x 512 B
        const int slice = std::max(256, vg_sw(SW_VG_BATCH_SLICE, 4096));
        int rc = VG_OK;
        const size_t qbytes = (size_t)c->dim * c->es;
        for (int q0 = 0; q0 < nq && rc == VG_OK; q0 += slice) {
            const int nqs = std::min(slice, nq - q0);
            rc = scan_topk_batch_mfma(c, metric, (const uint8_t *)queries + (size_t)q0 * qbytes, nqs, k, out_keys + (size_t)q0 * k,
                                      out_counts + q0);
        }
        if (rc != -1) {
            const bool quantized = (c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8);
            c->last_batch_path = quantized ? 2 : ((c->vtype != VG_TYPE_F32 || c->last_batch_half) ? 3 : 1);
            return rc;
        }
        for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    }
    // shapes the matrix-core kernels do not serve (f16 / bf16, L1, k > 32, rows > 512 floats / 1 KiB): the multi-query
    // scan (vg_scan_multi_kernel: 4 - or 2 for f16 / bf16 - queries share every row load of the HBM-bound pass) ...
    if (!few && k <= 64 && nq >= 2 && vg_sw(SW_VG_MULTI_SCAN, 1)) {
        int rc = scan_topk_batch_multi(c, metric, queries, nq, k, out_keys, out_counts);
        if (rc != -1) { c->last_batch_path = 5; return rc; }
        for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    }
    // ... or, when that has no kernel for the shape either (very long rows, k > 64), nq single-query scans
    c->last_batch_path = 6;
    const uint8_t *q = (const uint8_t *)queries;
    for (int i = 0; i < nq; ++i) {
        int rc = vg_scan_topk_keys(c, metric, q + (size_t)i * c->dim * c->es, k, out_keys + (size_t)i * k, out_counts + i);
        if (rc != VG_OK) return rc;
    }
    return VG_OK;
}

// Does a batch over this corpus return, for every (query, row) pair, the very float the single-query scan computes?  tie_order =
// reference looks for ties in a batch's keys and is defined on the single scan's arithmetic: integer sums are exact, f16 / bf16
// survivors carry the single scan's f64 arithmetic, f32 rows through the bf16 filter are re-evaluated with the single scan's f32
// chain - but the f32 matrix-core kernel (a k-ordered fmaf chain) and the multi-query scan (another launch shape) sum in another
// order, and a tie under the single scan's arithmetic could go unnoticed in their keys.
bool vg_batch_keys_are_scan_exact(const vg_corpus *c, int metric, int k) {
    if (c->vtype != VG_TYPE_F32) return true;
    if (!batch_f32_filter_eligible(c, metric, k)) return false;
    return vg_batch_lds_bytes(c->stride, k) == 0 || (batch_f32_filter_short_rows(c) && !c->filter_disabled);
}

extern "C" int vg_batch_last_path(const vg_corpus *c) { return c ? c->last_batch_path : 0; }
extern "C" int vg_batch_q8_status(const vg_corpus *c) { return c ? c->bq8_status : 0; }

extern "C" int vg_batch_filter_exact_evals(vg_corpus *c, unsigned long long *out_evals) {
    if (!c || !out_evals) return vg_fail(VG_ERR_INVALID, "vg_batch_filter_exact_evals: NULL argument");
    *out_evals = 0;
    if (!c->d_filter_evals) return VG_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long now = 0;
    HIP_TRY(hipMemcpy(&now, c->d_filter_evals + 1, sizeof(now), hipMemcpyDeviceToHost));
    *out_evals = now - c->bfilter_evals_read;
    c->bfilter_evals_read = now;
    return VG_OK;
}

extern "C" int vg_scan_topk_batch(vg_corpus *c, int metric, const void *queries, int nq, int k, int64_t *out_rowids,
                                  double *out_dist, int *out_counts) {
    if (!c || !queries || !out_counts) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_batch: NULL argument");
    if (nq <= 0) return VG_OK;
    for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_rowids || !out_dist) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_batch: NULL output");
    if (c->tie_order == VG_TIE_REFERENCE) {
        // The reference's order differs from (distance, position) only where equal distances meet among a query's k + 1 best
        // (vg_scan_topk_reference): the batch runs as it is - matrix cores included - with one more list slot, and only the
        // queries whose lists hold a tie are answered again, one by one, through the replaying scan.
        const size_t qbytes = (size_t)c->dim * c->es;
        const int k1 = k + 1;
        std::vector<uint64_t> keys1((size_t)nq * k1);
        std::vector<int> cnt1((size_t)nq, 0);
        // (k = 64: no slot to look for a tie with; f32 batches off the bf16 filter: not the single scan's floats - both query by query)
        int rcb = (k1 <= 64 && vg_batch_keys_are_scan_exact(c, metric, k1)) ? vg_scan_topk_batch_keys(c, metric, queries, nq, k1, keys1.data(), cnt1.data())
                                                                             : VG_ERR_UNSUPPORTED;
        if (rcb != VG_OK && rcb != VG_ERR_UNSUPPORTED) return rcb;
        for (int i = 0; i < nq; ++i) {
            bool tie = (rcb != VG_OK);
            const uint64_t *kq = &keys1[(size_t)i * k1];
            for (int j = 1; j < cnt1[(size_t)i] && !tie; ++j) tie = (kq[j] >> 32) == (kq[j - 1] >> 32);
            if (tie) {
                int rc1 = vg_scan_topk_reference(c, metric, (const uint8_t *)queries + (size_t)i * qbytes, k, out_rowids + (size_t)i * k,
                                                 out_dist + (size_t)i * k, &out_counts[i]);
                if (rc1 != VG_OK) return rc1;
                continue;
            }
            const int take = std::min(cnt1[(size_t)i], k);
            for (int j = 0; j < take; ++j) {
                out_dist[(size_t)i * k + j] = (double)vg_key_distance(kq[j]);
                out_rowids[(size_t)i * k + j] = vg_corpus_rowid_at(c, (int64_t)vg_key_position(kq[j]));
            }
            out_counts[i] = take;
        }
        return VG_OK;
    }
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
