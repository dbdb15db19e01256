/*
 * vectorgpu.h - C-ABI of the MI355X (gfx950) brute-force distance-scan + top-k engine.
 *
 * This is the drop-in boundary for sqlite-vector's hot path.  The SQLite extension host
 * (sqlite-vector_amd/ext/vector_ext.c, plain C) keeps the reference's SQL surface and calls only the
 * functions below; everything behind them is HIP.  Plain pointers and sizes, opaque handles, int return
 * codes (0 = VG_OK); no C++/torch/SQLite types.  Citations are file:line under /root/reference/src/.
 *
 * What each entry point replaces in the reference:
 *   vg_backend_name         distance_backend_name               distance-cpu.c:20, vector_backend() sqlite-vector.c:2549
 *   vg_corpus_*             the per-query "SELECT pk, col FROM tbl" row stream      sqlite-vector.c:2077-2096
 *                           and t_ctx->preloaded / precounter (quantized preload)   sqlite-vector.c:138-139, 1338-1404
 *   vg_scan_topk            vcursor_run_callback + vcursor_sort_callback            sqlite-vector.c:183-184
 *                           = vFullScanRun / vQuantRunMemory + vFullScanSortSlots   sqlite-vector.c:2071-2157, 2051-2069
 *                           incl. dispatch_distance_table[metric][type]             distance-cpu.c:21, distance-cpu.h:60
 *                           and the nearly_zero_float32 clamp                       sqlite-vector.c:994-996, 2099
 *   vg_scan_distances       the per-row body of the *_stream cursors                sqlite-vector.c:1901-1998
 *   vg_scan_topk_batch      (no reference entry point: Q independent vector_full_scan calls)
 *   vg_quantize_query       quantize_float32/16/b16/u8/i8 on the query              sqlite-vector.c:2166-2177, 495-757
 *
 * Result contract (differs from the reference only where the reference is history-dependent):
 *   - distances are the reference's float values (int8/uint8: bit-exact with distance-avx2.c; f32: <= 1e-5 rel);
 *   - order is ascending (distance, scan position); NaN and +Inf distances never enter (strict '<' vs INFINITY
 *     slots, sqlite-vector.c:1809,2102); fewer than k rows come back when fewer qualify (:1816-1817).
 */
#ifndef VECTORGPU_H
#define VECTORGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as the reference's enums, distance-cpu.h:36-58 */
enum { VG_TYPE_F32 = 1, VG_TYPE_F16 = 2, VG_TYPE_BF16 = 3, VG_TYPE_U8 = 4, VG_TYPE_I8 = 5 };
enum { VG_DIST_L2 = 1, VG_DIST_SQUARED_L2 = 2, VG_DIST_COSINE = 3, VG_DIST_DOT = 4, VG_DIST_L1 = 5 };
enum { VG_QUANT_AUTO = 0, VG_QUANT_U8 = 1, VG_QUANT_S8 = 2 };

enum {
    VG_OK = 0,
    VG_ERR_INVALID = 1,      /* bad argument */
    VG_ERR_NO_DEVICE = 2,    /* no usable gfx950 device / HIP runtime failure at init */
    VG_ERR_NOMEM = 3,        /* device or host allocation failed */
    VG_ERR_HIP = 4,          /* any other HIP runtime error (see vg_last_error) */
    VG_ERR_UNSUPPORTED = 5   /* valid request the engine does not implement (see vg_last_error) */
};

typedef struct vg_corpus vg_corpus;    /* one HBM-resident N x D matrix + host rowid map, on one device */
/* Threading: different handles may be used from different threads at the same time (each owns its stream and
 * buffers; vg_last_error is thread-local).  ONE handle must not be used by two threads at once - the extension
 * keeps one set of handles per connection, like the reference keeps one context per connection. */

/* ---- process / device ---- */
int         vg_device_count(void);                 /* number of visible HIP devices (0 if none) */
const char *vg_backend_name(void);                 /* "HIP gfx950" style string, static storage */
const char *vg_last_error(void);                   /* thread-local message of the last failing call */

/* ---- corpus staging ("stage once into HBM") ----
 * Rows are stored row-major with a 16-byte-multiple stride (zero padded), base 256-byte aligned, in scan order.
 * rowids == NULL means rowid = 1-based scan position (+ vg_corpus_set_rowid_base). */
int     vg_corpus_create(int device, int vtype, int dim, int64_t capacity_rows_hint, vg_corpus **out);
void    vg_corpus_destroy(vg_corpus *c);
int     vg_corpus_clear(vg_corpus *c);
int     vg_corpus_reserve(vg_corpus *c, int64_t capacity_rows);   /* grow HBM to hold that many rows (no-op if it does) */
int     vg_corpus_trim(vg_corpus *c);                              /* give back a reservation more than 25 % above the rows held */
int     vg_corpus_clone(const vg_corpus *src, vg_corpus **out);    /* the same rows, rowids and switches again (device-to-device copy): copy-on-write */
int64_t vg_corpus_rows(const vg_corpus *c);
int     vg_corpus_dim(const vg_corpus *c);
int     vg_corpus_type(const vg_corpus *c);
int     vg_corpus_device(const vg_corpus *c);
int64_t vg_corpus_hbm_bytes(const vg_corpus *c);   /* bytes of HBM held by the row matrix */
int     vg_corpus_set_rowid_base(vg_corpus *c, int64_t base);   /* implicit rowid = base + position (default 1) */

/* host rows, any byte stride >= dim*elem_size (NULL rows in the SQL table are simply not appended, :2093).
 * The rows are copied into a pinned bounce buffer before the call returns (the caller may reuse host_rows at once);
 * the H2D transfer itself is only enqueued, so staging overlaps with the caller producing the next block. */
int vg_corpus_append(vg_corpus *c, const void *host_rows, int64_t n_rows, int64_t row_stride_bytes,
                     const int64_t *rowids);
/* the reference's persisted/preloaded quantized format: n records of [int64 LE rowid][dim bytes], stride 8+dim
 * (sqlite-vector.c:1296-1309, 2127-2147).  Corpus type must be U8 or I8.  De-interleaved on the GPU. */
int vg_corpus_append_records(vg_corpus *c, const void *host_records, int64_t n_records);
/* Row maintenance - what keeps a resident corpus equal to the table after UPDATE / DELETE without re-reading the table (the
 * reference re-reads it for every scan, sqlite-vector.c:2077-2107).  vg_corpus_find_rowid: scan position of a rowid (-1: not
 * held; -2: the rowids were not appended in ascending order, no lookup).  vg_corpus_patch_rows: overwrite the rows at
 * `positions` with host_rows[i].  vg_corpus_delete_rows: take the rows at `positions` (strictly ascending) out and close the
 * gaps on the device - scan order, hence every tie-break, stays that of a fresh staging.  Per-row derived data (norms, shadow
 * copies) is re-made from the first touched row on by the next scan that needs it. */
int64_t vg_corpus_find_rowid(const vg_corpus *c, int64_t rowid);
int     vg_corpus_patch_rows(vg_corpus *c, const int64_t *positions, int64_t n, const void *host_rows, int64_t row_stride_bytes);
int     vg_corpus_delete_rows(vg_corpus *c, const int64_t *positions, int64_t n);
/* rows already in device memory (same device), e.g. produced by another kernel / a torch tensor */
int vg_corpus_append_device(vg_corpus *c, const void *dev_rows, int64_t n_rows, int64_t row_stride_bytes,
                            const int64_t *host_rowids);

/* ---- the hot path ---- */
/* One query (dim elements of the corpus type, host memory) -> the k best rows.
 * out_rowids[k], out_dist[k] (float values widened to double, like vFullScanCursor.distance), *out_count <= k.
 * Large f32 / f16 / bf16 corpora, k <= 64 (L2 / SQUARED_L2 / DOT / COSINE; f16 / bf16 also L1): the scan goes through a
 * provable LOWER BOUND of every row's distance and evaluates only the candidates exactly, with the plain kernel's own
 * arithmetic - the same rowids and distance bits as the plain scan.  From 2^20 rows and 512 MB it streams an int8 shadow copy
 * (per row: int8 elements + scale + residual norm; built on the first such scan after rows were appended or patched; + 26 %
 * device memory for f32, + 52 % for f16 / bf16; a quarter resp. half of the bytes per query); L1 (f16 / bf16) streams the rows
 * themselves (the bound replaces the reference's f64 chain for non-candidates).  VG_SCAN_FILTER_SHADOW=bf16: f32 corpora
 * (>= 3 GB) through a bf16 shadow copy (+ 50 %), f16 / bf16 (>= 1 GB) through their own rows.  vg_corpus_set_scan_filter(c, 0)
 * / VG_SCAN_FILTER=0 turn it off; a corpus whose shadow copy does not fit device memory, or whose rows the bound cannot tell
 * apart (it evaluates more than 1/8 of them), keeps the plain scan by itself.  uint8 / int8 corpora (2^20 rows, 768 MB): a
 * high-nibble shadow copy (+ 52 %, half the bytes), used only if a probe over a 2M-row prefix finds the data selective under it
 * (vg_corpus_set_scan_filter(c, 1): without the probe). */
int vg_scan_topk(vg_corpus *c, int metric, const void *query, int k,
                 int64_t *out_rowids, double *out_dist, int *out_count);

/* Same scan, but the k candidates stay on the device as packed 64-bit keys
 * (hi 32 = order-preserving image of the float distance, lo 32 = scan position), ascending, padded with
 * VG_KEY_EMPTY.  dev_query / dev_out_keys are device pointers (dev_query: the dim elements followed by ZERO bytes up
 * to the next 16-byte multiple; dev_out_keys: 64 keys); stream is a hipStream_t (NULL = the corpus' own stream).
 * No host synchronisation: this is what one rank of a row-sharded multi-GPU scan runs before the candidate
 * gather (RCCL) and what bench.py times. */
int vg_scan_topk_device(vg_corpus *c, int metric, const void *dev_query, int k,
                        uint64_t *dev_out_keys, void *stream);
#define VG_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
/* decode / merge helpers for gathered keys (host side, tiny): */
float   vg_key_distance(uint64_t key);
uint32_t vg_key_position(uint64_t key);
/* k-way merge of n_lists key lists (each list_len long, ascending, VG_KEY_EMPTY padded), list i's positions are
 * offset by pos_offsets[i] (NULL = 0) to form global scan positions; ties broken by (list index, position) which
 * equals global scan position order for row-range shards.  Returns count written (<= k). */
int vg_merge_keys(const uint64_t *keys, int n_lists, int list_len, const int64_t *pos_offsets, int k,
                  int64_t *out_global_pos, double *out_dist);

/* the same merge for nq queries at once: keys[list][query][list_len] (an all_gather of every shard's
 * vg_scan_topk_batch_keys output), out_global_pos / out_dist are nq x k, out_counts nq.  Returns 0, -1 on bad arguments. */
int vg_merge_keys_batch(const uint64_t *keys, int n_lists, int nq, int list_len, const int64_t *pos_offsets, int k,
                        int64_t *out_global_pos, double *out_dist, int *out_counts);

/* The host-side scan again, but returning the packed keys (positions, not rowids) - the form a multi-shard caller
 * merges.  vg_scan_topk_keys: any k, out_keys[min(k, rows)], synchronous.  enqueue / collect: the fused path
 * (k <= 64) split in two, so that several corpora (one per device) are all in flight before the first wait;
 * collect fills 64 keys (VG_KEY_EMPTY padded; all empty for an empty corpus). */
int vg_scan_topk_keys(vg_corpus *c, int metric, const void *query, int k, uint64_t *out_keys, int *out_count);
int vg_scan_topk_enqueue(vg_corpus *c, int metric, const void *query, int k);
int vg_scan_topk_collect(vg_corpus *c, uint64_t *out_keys64);

/* All N distances in scan order (clamp applied), for the *_stream table-valued functions. */
int vg_scan_distances(vg_corpus *c, int metric, const void *query, float *out_dist_host);
int vg_scan_distances_device(vg_corpus *c, int metric, const void *dev_query, float *dev_out_dist, void *stream);
/* rowid of a scan position (host map) */
int64_t vg_corpus_rowid_at(const vg_corpus *c, int64_t position);

/* nq queries at once (row-major nq x dim, host).  out_rowids / out_dist are nq x k, out_counts nq.
 * f32 corpora, k <= 32, rows <= 512 floats, metric DOT / COSINE / L2 / SQUARED_L2: one pass over the corpus on the
 * matrix cores (Q x C^T tiles feed per-query candidate lists; L2 survivors are re-evaluated with the direct formula);
 * rows of 513 .. 1024 floats: a bf16 shadow copy of the corpus feeds the matrix cores as a filter, every candidate is
 * re-evaluated on the f32 rows with the single scan's arithmetic - and so do shorter rows of a corpus the filter scan serves
 * (switched on, >= 3 GB: the shadow copy those scans make; the GEMM at the bf16 rate over half the bytes, the distances are
 * the single scans' distances; a selectivity guard falls back to the f32 matrix-core kernel on data the bound does not
 * separate; environment VG_F32_FILTER=0 / 1 forces it off / on).
 * uint8 / int8 corpora, k <= 32, rows <= 2048 bytes, same metrics: the integer matrix cores, results identical to
 * nq vg_scan_topk calls.  f16 / bf16 corpora, k <= 32, rows <= 1024 elements, same metrics: the matrix cores filter,
 * every candidate is re-evaluated with the single scan's f64 arithmetic.  Other shapes on f32 / uint8 / int8 corpora
 * with k <= 64: 4 (long rows: 2) queries share each pass of the scan kernel.  Everything else: nq single-query
 * scans.  Same result contract as vg_scan_topk (float batches: tolerance 1e-5, quantized batches bit-exact). */
int vg_scan_topk_batch(vg_corpus *c, int metric, const void *queries, int nq, int k,
                       int64_t *out_rowids, double *out_dist, int *out_counts);

/* same, as keys: out_keys nq x k (positions local to this corpus) */
int vg_scan_topk_batch_keys(vg_corpus *c, int metric, const void *queries, int nq, int k,
                            uint64_t *out_keys, int *out_counts);

/* ---- query-side quantizer (host, bit-exact with sqlite-vector.c:495-757) ---- */
int vg_quantize_query(int src_type, const void *src, int dim, float scale, float offset, int qtype, void *dst);

/* ---- vector_quantize on the staged corpus (sqlite-vector.c:1147-1336 moved to the GPU) ----
 * pass 1: global min / max / any-negative over every element, widened to float like :1228-1254;
 * pass 2: rows [row0, row0+n) quantized with (scale, offset, qtype) into n x dim tightly packed bytes on the host,
 *         bit-exact with quantize_* (:495-757).  The caller interleaves rowids into the persisted record format. */
int vg_corpus_minmax(vg_corpus *c, float *out_min, float *out_max, int *out_any_negative);
int vg_corpus_quantize_rows(vg_corpus *c, float scale, float offset, int qtype, int64_t row0, int64_t n_rows,
                            uint8_t *out_host);

/* ---- one logical corpus over several devices of ONE process (vg_shards.hip) ----
 * What the SQLite extension holds per (table, column): a connection lives in one process, so its multi-GPU form
 * is S per-device corpora driven from one host thread, not S processes.  Rows are dealt out
 * block-cyclically in scan order (block_rows rows per block, 0 = 65536); every call below is the vg_corpus_* /
 * vg_scan_* call of the same name with results merged by (distance, GLOBAL scan position): bit-identical to a
 * single corpus holding all rows.  devices == NULL means devices 0..n-1; the same device may be listed more than
 * once (logical shards - how the tests exercise this on a 1-GPU box).  With n_devices == 1 every call forwards. */
typedef struct vg_shards vg_shards;
int     vg_shards_create(const int *devices, int n_devices, int vtype, int dim, int64_t block_rows, vg_shards **out);
void    vg_shards_destroy(vg_shards *s);
int     vg_shards_clear(vg_shards *s);
int     vg_shards_count(const vg_shards *s);
int64_t vg_shards_rows(const vg_shards *s);
vg_corpus *vg_shards_shard(const vg_shards *s, int i);          /* borrow shard i (profiling, introspection) */
int     vg_shards_reserve(vg_shards *s, int64_t total_rows);
int     vg_shards_trim(vg_shards *s);                                                      /* vg_corpus_trim on every shard */
int     vg_shards_clone(const vg_shards *src, vg_shards **out);                            /* vg_corpus_clone of every shard */
int     vg_shards_set_rowid_base(vg_shards *s, int64_t base);
int     vg_shards_append(vg_shards *s, const void *host_rows, int64_t n_rows, int64_t row_stride_bytes, const int64_t *rowids);
int     vg_shards_append_records(vg_shards *s, const void *host_records, int64_t n_records);
int64_t vg_shards_rowid_at(const vg_shards *s, int64_t position);
int     vg_shards_rowids(const vg_shards *s, int64_t pos0, int64_t n, int64_t *out);   /* rowids of positions [pos0, pos0 + n) */
int64_t vg_shards_find_rowid(const vg_shards *s, int64_t rowid);                       /* single-shard handles only (else -2 / VG_ERR_UNSUPPORTED): */
int     vg_shards_patch_rows(vg_shards *s, const int64_t *positions, int64_t n, const void *host_rows, int64_t row_stride_bytes);
int     vg_shards_delete_rows(vg_shards *s, const int64_t *positions, int64_t n);
int     vg_shards_set_scan_filter(vg_shards *s, int mode);                             /* vg_corpus_set_scan_filter on every shard */
/* How the S x 64 candidate keys of a top-k scan reach the host: 0 = every shard copies its own 64 keys back (default), 1 = ONE grouped
 * ncclAllGather over RCCL / xGMI on the scan streams + one copy from the first device (librccl.so is dlopen'ed on first use; devices
 * must be distinct; any failure falls back to 0 for the handle).  Default from VECTORGPU_SHARD_GATHER=host|rccl.  Same keys, same
 * merge, same result either way (vg_shards_gather_stats, vectorgpu_diag.h, counts which form served). */
int     vg_shards_set_gather(vg_shards *s, int mode);
int     vg_shards_scan_topk(vg_shards *s, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist, int *out_count);
int     vg_shards_scan_topk_batch(vg_shards *s, int metric, const void *queries, int nq, int k,
                                  int64_t *out_rowids, double *out_dist, int *out_counts);
int     vg_shards_scan_distances(vg_shards *s, int metric, const void *query, float *out_dist_host);
int     vg_shards_minmax(vg_shards *s, float *out_min, float *out_max, int *out_any_negative);
int     vg_shards_quantize_rows(vg_shards *s, float scale, float offset, int qtype, int64_t row0, int64_t n_rows, uint8_t *out_host);

/* Diagnostics - profiling switches, kernel timings, launch plans, counters and the building blocks the tests compose - are declared
 * in vectorgpu_diag.h: exported by the same library, not part of the surface a binding has to keep stable. */

/* ---- tie order ----
 * VG_TIE_POSITION (default): results ordered by (distance, scan position) - deterministic, shard-invariant.
 * VG_TIE_REFERENCE: the reference's own result among EQUAL distances, rowid for rowid: its k unsorted slots, strict '<'
 * insertion into the FIRST slot holding the maximum and the final exchange sort (sqlite-vector.c:2022-2069, 2102-2106,
 * 2138-2146) are replayed on the host over the rows that can enter at all - the first P rows, then only the rows the
 * device finds below the bound reached so far (vg_reforder.hip).  Same distances either way; this mode makes
 * int8 / uint8 scans (where ties are routine) identical to the reference in rowids and order too.  With the mode set,
 * vg_scan_topk / vg_shards_scan_topk answer through vg_scan_topk_reference (k <= 64: the ordinary scan with one more list slot, a
 * host replay over what the scan left behind only when the k + 1 best hold a tie - no second scan, on any number of shards); the
 * batch calls run with one more slot too and answer only the queries with a tie again. */
enum { VG_TIE_POSITION = 0, VG_TIE_REFERENCE = 1 };
int vg_corpus_set_tie_order(vg_corpus *c, int mode);
int vg_corpus_tie_order(const vg_corpus *c);
int vg_shards_set_tie_order(vg_shards *s, int mode);
int vg_scan_topk_reference(vg_corpus *c, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist, int *out_count);
/* ---- out-of-core scans: a table that does not fit device memory ----
 * The reference scans a table of ANY size: vFullScanRun walks the statement's rows one by one (sqlite-vector.c:2071-2113) and keeps k
 * slots.  Its form here: the caller hands the rows over in scan order, the device holds TWO slabs of slab_rows rows each - one being
 * filled through the pinned bounce buffers, the other one being scanned on a host thread of its own - and what is carried from slab to
 * slab is what the reference carries from row to row: the k best so far (tie_order = position: merged by (distance, scan position);
 * tie_order = reference: the slot state itself, offered the rows of each later slab that lie below the bound reached, vg_refslots.h).
 * Results are those of vg_scan_topk over one corpus holding all rows, bit for bit.  k = 0: every row's distance and rowid instead
 * (the *_stream functions), fetched with vg_slab_scan_all.  rowid_base: implicit rowid of scan position p = rowid_base + p where the
 * caller passes rowids = NULL.  The handle serves one query; the extension makes one per scan of a table it cannot keep resident
 * (VECTORGPU_HBM_LIMIT or the device's free memory: INTEGRATION.md). */
/* Pinned host memory for a HOST-RESIDENT copy of a table that does not fit the device (round 6): the tier between "staged in HBM" and the
 * reference's own cost model of reading the table for every query (sqlite-vector.c:2077-2107).  Rows handed to vg_slab_scan_rows /
 * vg_corpus_append from such a block are read by the DMA engine where they lie (no bounce copy): a scan of the copy runs at the host
 * link's rate.  VG_ERR_NOMEM when the runtime refuses to pin that much. */
int  vg_host_alloc(size_t bytes, void **out);
void vg_host_free(void *p);

typedef struct vg_slab_scan vg_slab_scan;
int  vg_slab_scan_begin(int device, int vtype, int dim, int metric, const void *query, int k, int tie_order, int64_t slab_rows,
                        int64_t rowid_base, vg_slab_scan **out);
int  vg_slab_scan_rows(vg_slab_scan *s, const void *host_rows, int64_t n_rows, int64_t row_stride_bytes, const int64_t *rowids);
int  vg_slab_scan_records(vg_slab_scan *s, const void *host_records, int64_t n_records);   /* [int64 LE rowid | dim bytes] records (uint8 / int8) */
int  vg_slab_scan_finish(vg_slab_scan *s, int64_t *out_rowids, double *out_dist, int *out_count);
int  vg_slab_scan_all(vg_slab_scan *s, int64_t *out_rows, const float **out_dist, const int64_t **out_rowids);   /* k = 0, after finish; valid until destroy */
void vg_slab_scan_destroy(vg_slab_scan *s);
int  vg_device_memory(int device, long long *out_free_bytes, long long *out_total_bytes);

/* ---- per-corpus switches ---- */
/* The lower-bound filter scans of single top-k queries (see vg_scan_topk): 0 = off (plain scans, no shadow copy is built),
 * -1 = default (environment VG_SCAN_FILTER, else on): f32 / f16 / bf16 corpora from 2^20 rows and 512 MB through an int8 shadow
 * copy (+ 26 % / + 52 % device memory), uint8 / int8 corpora through a high-nibble copy (+ 52 %) only after a probe of a 2M-row
 * prefix found the data selective under it; 1 = on where it serves WITHOUT that probe (uint8 / int8: the + 52 % is spent at
 * once).  The extension maps vector_init's scan_filter= option here - for the raw-vector corpus and the quantized one alike. */
int vg_corpus_set_scan_filter(vg_corpus *c, int mode);

#ifdef __cplusplus
}
#endif
#endif
