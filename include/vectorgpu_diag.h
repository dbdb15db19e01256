/*
 * vectorgpu_diag.h - diagnostics of libvectorgpu.so: profiling switches, kernel timings, launch plans, counters and the building
 * blocks tests / bench.py / tools compose.  Exported by the same library as vectorgpu.h, but NOT part of the drop-in boundary: a
 * binding of the hot path (the SQLite extension host) needs none of these and they may change between builds.
 */
#ifndef VECTORGPU_DIAG_H
#define VECTORGPU_DIAG_H

#include "vectorgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- instrumentation ---- */
/* When enabled, every scan records HIP events around its kernels on the stream they run on (a ring of 1024
 * launches, no host synchronisation at launch time).  Enabling resets the launch counter. */
int vg_set_profiling(vg_corpus *c, int enabled);
/* milliseconds of the last scan's kernels: scan (dominant, HBM-bound) and merge (candidate reduction). */
int vg_last_kernel_ms(vg_corpus *c, float *scan_ms, float *merge_ms);
/* mean kernel milliseconds over the launches recorded since profiling was enabled (at most the last 1024) */
int vg_profile_mean_ms(vg_corpus *c, int *n_launches, float *scan_ms, float *merge_ms);
/* the same, with the filter scan's plain-f32 pre-pass (its scan + merge, run before the filter kernel) reported on its
 * own: scan_ms is ONE kernel - the dominant one - whichever path served the scan */
int vg_profile_mean_ms_ex(vg_corpus *c, int *n_launches, float *scan_ms, float *merge_ms, float *prepass_ms);
/* the launch shape the plain scan kernel would run a row of `dim` elements of `vtype` with under `metric`: lanes sharing a row,
 * 16-byte chunks per lane, and whether the row takes the long-row kernel instead (pure host logic - no device, no corpus) */
int vg_plan_scan_shape(int vtype, int dim, int metric, int *lanes_per_row, int *chunks_per_lane, int *long_rows);
/* the form a batch of nq queries over f16 / bf16 rows (or the bf16 shadow rows of an f32 corpus) of stride_bytes runs in: wavefronts
 * per workgroup (32 queries each) and workgroups per CU - one of eight, or two of four where both fit the CU's LDS.  Host logic
 * only; returns -1 for rows the matrix-core kernel does not serve */
int vg_batch_h_plan(long long stride_bytes, int k, int nq, int *waves, int *blocks_per_cu);
/* name of the scan kernel variant chosen for (metric) on this corpus, e.g. "scan_f32_l2_u6_lpr16" */
const char *vg_scan_kernel_name(vg_corpus *c, int metric);
/* filter scan: f32 rows evaluated exactly by the filter-scan launches since the last call (then reset) - how selective the
 * bf16 bound is on the data at hand */
int vg_filter_exact_evals(vg_corpus *c, unsigned long long *out_evals);
/* the filter scan's selectivity guard: > 0 = the next that many single scans of this corpus take the PLAIN kernel (the last filter
 * launches evaluated more than 1/8 of the rows exactly: data the bound does not separate), 0 = the filter scan is in use */
int vg_filter_guard_cooldown(vg_corpus *c);
/* the same for f32 batches through the bf16 filter (vg_scan_topk_batch): (query, row) pairs evaluated exactly since the last call */
int vg_batch_filter_exact_evals(vg_corpus *c, unsigned long long *out_evals);
/* which path answered this corpus' last vg_scan_topk_batch[_keys] call: 0 none yet, 1 the f32 matrix-core kernel, 2 the int8 one,
 * 3 the half-precision kernel (f16 / bf16 rows, f32 rows through their bf16 shadow copy), 4 the long-row form of it (1025 .. 3072
 * elements: the K dimension split over a workgroup's wavefronts), 5 the multi-query scan, 6 one scan per query, 7 the int8 filter over
 * an f32 corpus' int8 shadow copy (vg_batch_q8.hip: batches of more than 256 queries over corpora of 2^20+ rows) */
/* The engine reads its environment switches (VG_*, csrc/vg_switches.h) when the library is loaded and whenever a corpus / a shard set is
 * created - never on a query's path.  A test that changes one between two calls on an existing corpus re-reads them with this. */
void vg_reload_switches(void);
int vg_batch_last_path(const vg_corpus *c);
/* how this corpus' last attempt at the int8 batch filter ended: 0 it answered, 1 no room for the tile-major int8 copy, 2 shape not
 * served, 3 a pair region overflowed (the batch was answered by another path and the next 16 batches skip the int8 filter) */
int vg_batch_q8_status(const vg_corpus *c);

/* kernel milliseconds (HIP events on the corpus stream) and rows of the corpus' last vg_corpus_minmax (which = 0) /
 * vg_corpus_quantize_rows (1) pass, and - while profiling is on - of the last int8 shadow-copy pass of the filter scans (2) */
int vg_corpus_pass_ms(const vg_corpus *c, int which, float *out_ms, long long *out_rows);
/* rows sent to a device by every vg_corpus_append* / vg_shards_append* call of this process so far */
long long vg_stat_rows_appended(void);

/* ---- counters ---- */
/* queries of a shards handle served by [0] the host gather, [1] the RCCL all-gather; returns 1 while RCCL serves */
int vg_shards_gather_stats(vg_shards *s, unsigned long long *out2);
/* 1 when every query's per-shard work (enqueue + collect) runs on the handle's persistent host threads, one per shard: the default
 * for shards on more than one device, VECTORGPU_SHARD_THREADS=1 / 0 forces it (vg_shards.hip: pool_run) */
int vg_shards_threaded(const vg_shards *s);
/* counters of the reference-order scans of this corpus / shards handle so far: [0] scans, [1] scans whose k + 1 best distances held a
 * tie (the others cost what a tie_order = position scan costs; k = 64 has no 65th slot and always counts), [2] of those, answered by
 * the fused replay (prefix pass + the candidates the scan emitted: no second pass over the corpus), [3] answered by the store-mode
 * replay (k > 64, rows too long for an emitting kernel, candidate-stream overflow) */
int vg_corpus_tie_stats(const vg_corpus *c, unsigned long long *out4);
int vg_shards_tie_stats(const vg_shards *s, unsigned long long *out4);
/* bytes a corpus (all shards of a vg_shards handle) holds on its device(s), by allocation size: out3[0] the row matrix, [1] per-row data
 * derived from it (the filter scans' shadow copies, the batch kernels' tile-major copies, norms, row statistics), [2] working buffers.
 * The extension reports them through vector_gpu_memory(table, column). */
int vg_corpus_device_bytes(const vg_corpus *c, long long *out3);
int vg_shards_device_bytes(const vg_shards *s, long long *out3);
/* building blocks (what vg_shards composes over several devices): all N distances of a query left in device memory
 * (enqueued, no wait); rows [pos0, pos0 + n) of them; every row >= pos0 whose distance is < bound as
 * (position << 32 | float bits) pairs in any order - *out_count may exceed cap, then only cap pairs were written. */
int vg_scan_distances_resident(vg_corpus *c, int metric, const void *query);
int vg_resident_distances_fetch(vg_corpus *c, int64_t pos0, int64_t n, float *out_host);
int vg_resident_distances_below(vg_corpus *c, int64_t pos0, float bound, uint64_t *out_pairs, int64_t cap, int64_t *out_count);

/* host-only: the same replay over n distances the caller holds (scan order); returns the count (<= k) or -1.
 * below_cap <= 0: the device path's candidate capacity. */
int vg_reference_topk_replay(const float *dist, int64_t n, int k, int64_t below_cap, int64_t *out_pos, double *out_dist);
/* the same stream handed over slab by slab (the out-of-core scan's continuation of the slot state, host-only: CPU tests) */
int vg_reference_topk_replay_slabs(const float *dist, int64_t n, int k, int64_t slab_rows, int64_t below_cap, int64_t *out_pos, double *out_dist);


#ifdef __cplusplus
}
#endif
#endif
