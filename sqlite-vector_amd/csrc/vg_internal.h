// vg_internal.h - what the host-side translation units of libvectorgpu.so share: the corpus object, the error
// helpers and the few internal entry points that cross files.  Not part of the C-ABI (include/vectorgpu.h is).
//   vg_corpus.hip     corpus lifetime + staging, key / quantizer / instrumentation helpers
//   vg_api.hip        kernel selection, the scan launches (needs the kernel templates of vg_scan.h)
//   vg_batch_api.hip  batched queries: planning, cached row statistics, launches of vg_batch*.hip
#pragma once

#include "../../include/vectorgpu.h"
#include "../../include/vectorgpu_diag.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define VG_PROF_RING 1024
#define VG_PROF_EVS 4             // events per recorded launch: start | after the pre-pass | after the scan kernel | after the merge
#define VG_EVF_MERGE 1            // ev_flags bits: the launch had a merge kernel / a pre-pass of its own
#define VG_EVF_PREPASS 2
#define VG_WAVE_HOST 64

int vg_fail(int code, const char *fmt, ...);         // sets the thread-local message, returns code (vg_corpus.hip)

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return vg_fail(e__ == hipErrorOutOfMemory ? VG_ERR_NOMEM : VG_ERR_HIP, "%s failed: %s (%s:%d)", \
                           #expr, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)


static inline int vg_elem_size(int vtype) {
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
    bool rowids_ascending = true;  // every appended rowid was larger than the one before it (a rowid table in key order)
    int64_t rowid_base = 1;

    hipStream_t stream = nullptr;
    uint8_t *d_query = nullptr;    // nch*16 bytes
    uint8_t *h_query = nullptr;    // pinned
    uint64_t *d_cand = nullptr;    // max_blocks * 64 keys
    uint64_t *d_cand_pre = nullptr; // the same size: a filter scan's pre-pass lists, read unmerged by the filter kernel
    uint64_t *d_keys = nullptr;    // 64 keys
    uint64_t *h_keys = nullptr;    // pinned, 64 keys
    float *d_dist = nullptr;       // lazily sized to n_rows (stream scans / large k / tie_order = reference)
    int64_t d_dist_cap = 0;
    int64_t dist_valid_rows = 0;   // rows of d_dist the last vg_scan_distances_resident filled (0: none)
    unsigned long long *d_below = nullptr;   // vg_reforder.hip: [count | VG_BELOW_CAP (position, distance) pairs]
    uint8_t *h_ref = nullptr;      // pinned landing zone of its small device-to-host copies (a pageable destination costs ~100 us each)
    size_t h_ref_bytes = 0;
    std::vector<uint64_t> ref_pairs;   // its candidate pairs on the host (kept between scans: no 1 MB clear per query)
    int tie_order = 0;             // VG_TIE_POSITION / VG_TIE_REFERENCE (vg_corpus_set_tie_order)
    // tie_order = reference, the fused form (vg_reforder.hip): a top-k scan with one more list slot; only when its k+1 best distances
    // hold a tie does the host replay the reference's slots - over the prefix pass' distances + the candidate stream the scan emitted
    float *d_ref_prefix = nullptr; // distances of the first ref_prefix_rows rows (written by the prefix pass: top-k + store)
    int64_t ref_prefix_cap = 0;
    int64_t ref_prefix_rows = -1;  // rows the LAST emitting launch stored (-1: that launch could not emit - long rows / a small corpus)
    int ref_hot = 0;               // > 0: ties were seen recently - plain-kernel scans pay the prefix pass up front instead of a second scan
    unsigned long long ref_stats[4] = {0, 0, 0, 0};   // reference-order scans | with a tie among the k+1 best | answered by the fused replay | by the store-mode replay
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
    uint32_t *d_sx = nullptr;                     // per row: (sum x, sum x^2)
    uint8_t *d_rows_s8 = nullptr;
    uint8_t *d_rows_tm = nullptr;                 // f16 / bf16 corpora: the tile-major copy the batched matrix-core kernel streams
    int64_t tm_rows = 0, tm_cap = 0;
    bool tm_disabled = false;                     // (it did not fit: the kernel gathers from the row-major corpus)
    uint8_t *d_rows_bf = nullptr;                 // f32 corpora: bf16 shadow copy for the matrix-core filter (vg_batch_h.hip)
    int64_t bf_rows = 0, bf_cap = 0;
    uint8_t *d_rows_q8 = nullptr;                 // f32 corpora: int8 shadow copy for the single-query filter scan (vg_filter.hip) ...
    void *d_q8stat = nullptr;                     // ... and per row (scale, residual norm) as float2
    int64_t q8_rows = 0, q8_cap = 0;
    uint8_t *d_rows_q8tm = nullptr;               // f32 corpora: the TILE-MAJOR copy of that int8 shadow, what the int8 batch filter streams (vg_batch_q8.hip) ...
    void *d_q8tm_stat = nullptr;                  // ... and per row (scale | -1, residual norm, ||x||, 0) as float4, two tiles of slack behind the last row
    int64_t q8tm_rows = 0, q8tm_cap = 0;
    bool q8tm_disabled = false;                   // (it did not fit: batches keep the bf16 filter / the f32 matrix-core kernel)
    int bq8_cooldown = 0;                         // > 0: a batch overflowed a pair region - that many batches take the other paths
    int bq8_status = 0;                           // the last int8-filter batch: 0 answered, 1 no room for the copies, 2 shape not served, 3 a pair region overflowed (vg_batch_q8_status)
    uint8_t *d_rows_n4 = nullptr;                 // uint8 / int8 corpora: high-nibble shadow copy for the filter scan (vg_scan_filter_n4.h) ...
    void *d_n4stat = nullptr;                     // ... and per row (sum x^2, sum of low nibbles, their centred norm), 16 bytes
    int64_t n4_rows = 0, n4_cap = 0;
    bool n4_disabled = false;
    int n4_probe = 0;                             // 0 = not probed yet, 1 = selective on this data (filter on), 2 = not selective (plain kernel)
    int64_t n4_probe_rows = 0;                    // n_rows at that probe (an unselective corpus is probed again once it has doubled)
    bool q8_disabled = false;                     // (it did not fit next to an f16 / bf16 corpus: the filter scans read the rows themselves)
    bool filter_disabled = false;                 // the shadow copy / norms did not fit HBM: single queries keep the plain f32 scan
    int scan_filter_mode = -1;                    // vg_corpus_set_scan_filter: -1 = default (env VG_SCAN_FILTER, else on), 0 = off, 1 = on
    unsigned long long *d_filter_evals = nullptr; // filter scan: exact evaluations so far (device counter, one atomic per workgroup) ...
    unsigned long long *h_filter_evals = nullptr; // ... and its pinned host mirror, refreshed by an 8-byte copy behind every filter launch: the
                                                  //   host can look at it without a wait (system-scope atomics straight into host memory were
                                                  //   tried: 256 of them per launch serialise on the host link, +160 us per scan)
    unsigned long long filter_evals_read = 0;     // its value at the last vg_filter_exact_evals read-out
    unsigned long long filter_evals_seen = 0;     // ... and when the selectivity guard last looked
    long long filter_launches_seen = 0, filter_launches = 0;
    int filter_cooldown = 0;                      // > 0: the bound is not selective on this data - that many scans take the plain kernel
    int filter_prepass_div = 128;                 // the filter scan's pre-pass covers 1 / this of the rows (16 .. 128, follows the candidate rate)
    // f32 batches through the bf16 filter (vg_batch_api.hip): counter [1] of d_filter_evals, the same kind of guard
    unsigned long long bfilter_evals_seen = 0, bfilter_evals_read = 0;
    long long bfilter_pairs = 0;                  // (query, row) pairs of the filtered batches since the guard last looked
    int bfilter_cooldown = 0;                     // > 0: that many batches take the f32 matrix-core kernel
    int64_t i8_rows = 0, i8_cap = 0;              // orders a caller-stream scan behind a norm pass on the corpus stream
    uint8_t *h_bq = nullptr;       // long-row batches: the pinned buffer the padded queries go up through (+ their norms coming back)
    size_t h_bq_bytes = 0;
    void *d_bq = nullptr;          // batched path: padded queries, per-(query, partition) candidates, final keys
    uint64_t *d_bcand = nullptr, *d_bkeys = nullptr;
    size_t bq_bytes = 0, bcand_bytes = 0, bkeys_bytes = 0;
    uint64_t *d_bpairs = nullptr;  // half-precision batches, split form (vg_batch_h.hip): candidate pairs per filter wavefront ...
    uint32_t *d_bpcounts = nullptr; // ... and their counts + overflow flag
    size_t bpairs_bytes = 0, bpcount_bytes = 0;
    bool bsplit_off = false;       // a batch overflowed a pair region: this corpus keeps the fused kernel
    int last_batch_path = 0;       // vg_batch_last_path (vectorgpu_diag.h)
    bool last_batch_half = false;  // (the last matrix-core batch went through vg_batch_h.hip)
    int blong_cooldown = 0;        // ... of the long-row kernel (vg_batch_hl.hip): the next batches over this corpus take the multi-query scan
    int max_blocks = 0;
    int cu_count = 0;

    // instrumentation: a ring of event quadruples (start | after the pre-pass | after the scan kernel | after the merge),
    // recorded on the stream each launch runs on, read back only when asked - no host synchronisation inside a timed region
    bool profiling = false;
    std::vector<hipEvent_t> ev;            // VG_PROF_EVS * VG_PROF_RING events, created on first enable
    std::vector<uint8_t> ev_flags;         // VG_EVF_* per slot
    long long prof_launches = 0;           // launches recorded since profiling was (re)enabled
    float last_scan_ms = 0.f, last_merge_ms = 0.f, last_prepass_ms = 0.f;
    char kernel_name[64] = {0};
    // kernel milliseconds / rows of the last minmax [0], quantize [1] and int8-shadow [2] pass (HIP events on the corpus stream;
    // the first two always, the shadow pass while profiling is on) - bench.py --workload stage prices them against the HBM peak
    float pass_ms[3] = {0.f, 0.f, 0.f};
    long long pass_rows[3] = {0, 0, 0};
};

#include "vg_switches.h"            // the environment switches: vg_sw(SW_..., default)

// What one scan launch covers.  n_rows < 0: the whole corpus; a prefix otherwise (the filter scan's plain pre-pass).
struct ScanPlan {
    int64_t n_rows = -1;
    bool allow_filter = true;      // false: the plain kernel of vg_scan.h whatever the corpus / the switches say
    bool record = true;            // false: no entry in the profiling ring (a pre-pass is recorded by its parent launch)
    // tie_order = reference (vg_reforder.hip).  ref_emit: a whole-corpus top-k scan that also leaves behind what a host replay of the
    // reference's slot algorithm needs - the distances of the first c->ref_prefix_rows rows (c->d_ref_prefix, written by a prefix pass
    // that runs in front: top-k + store) and every later row that can enter the slots (c->d_below, emitted by the lists).
    bool ref_emit = false;
    float *store_prefix = nullptr;            // (the prefix pass itself) top-k mode that also stores every row's distance here ...
    unsigned long long *emit_reset = nullptr; // ... and zeroes the candidate counter of the pass behind it
    // a pre-pass whose per-CU lists are consumed UNMERGED by the kernel behind it (vg_kth_head): lists go to lists_out, no merge launch
    uint64_t *lists_out = nullptr;
    int *n_lists_out = nullptr;
    // where the LAST merge of the launch leaves the k winners, if not in dev_out_keys: the corpus' pinned, device-mapped h_keys -
    // the one workgroup of the merge writes its 512 bytes straight across the host link and the copy command behind it goes away
    // (anything in front that reads dev_out_keys on the device - a pre-pass' threshold keys - still finds them in device memory)
    uint64_t *final_out = nullptr;
};
#define VG_BELOW_CAP (1 << 17)        // candidate pairs the device buffer holds (more: the store-mode replay takes over)
#define VG_REF_EMIT_MIN_ROWS (1 << 17) // below this a reference-order scan with a tie simply replays a store-mode scan (cheap at that size)
#define VG_REF_PREFIX_MAX (1 << 20)   // rows of the prefix pass whose distances travel to the host on a tie (4 MB of pinned memory)
#define VG_REF_PINNED_BYTES (((size_t)VG_REF_PREFIX_MAX * 4) + ((size_t)VG_BELOW_CAP + 1) * 8)
static inline int64_t vg_ref_prefix_for(int64_t n_rows) {
    int64_t p = n_rows / 128;
    if (p < 16384) p = 16384;
    if (p > VG_REF_PREFIX_MAX) p = VG_REF_PREFIX_MAX;
    return p;
}
int vg_ensure_ref_buffers(vg_corpus *c, int64_t prefix_rows);       // vg_reforder.hip: d_ref_prefix / d_below / h_ref
#define VG_REF_FIRST_PAIRS 8191       // candidate pairs that travel with the counter in the first copy (nearly always all of them)
// vg_reforder.hip: what the last emitting launch of a corpus left behind, brought to the host (enqueue the copies, then wait)
int vg_ref_emitted_enqueue(vg_corpus *c);
int vg_ref_emitted_wait(vg_corpus *c, const float **prefix, int64_t *prefix_rows, unsigned long long **pairs, unsigned long long *count,
                        bool *overflow);
struct VgRefSlots;
void vg_ref_offer_run(VgRefSlots &slots, const float *d, int64_t n, int64_t g0);   // a run of consecutive rows offered to the slots
int vg_ref_replay_slab(vg_corpus *c, int metric, const void *query, int k, VgRefSlots &slots, int64_t gbase, bool fresh);   // vg_reforder.hip

// next slot of the profiling ring (nullptr when profiling is off)
static inline hipEvent_t *vg_prof_slot(vg_corpus *c, uint8_t flags) {
    if (!c->profiling) return nullptr;
    const int slot = (int)(c->prof_launches % VG_PROF_RING);
    c->ev_flags[(size_t)slot] = flags;
    ++c->prof_launches;
    return &c->ev[(size_t)slot * VG_PROF_EVS];
}

// ---- internal entry points that cross translation units
int vg_metric_to_acc(int metric);                                  // vg_api.hip; -1 for an unknown metric
void vg_collect_timing(vg_corpus *c);                              // vg_api.hip: event times of the last launch
int vg_ensure_row_norms(vg_corpus *c);                             // vg_api.hip (uses the f16 / bf16 norm kernel of vg_scan.h)
struct VgShape { int lpr_log2; int U; bool long_rows; };           // launch shape of a scan: lanes per row, chunks per lane
bool vg_choose_shape(int nch, int vtype, int acc, VgShape *out, int u_cap);   // vg_api.hip
int vg_launch_merge(const uint64_t *dev_cand, int nlists, int k, uint64_t *dev_out_keys, int nq, hipStream_t stream);   // vg_api.hip
// vg_api.hip <-> vg_filter.hip (the filter scans of vg_scan_filter.h are a translation unit of their own)
int vg_launch_plain_scan(vg_corpus *c, int metric, const uint8_t *dev_query, int k, uint64_t *dev_out_keys, hipStream_t stream,
                         const ScanPlan &plan);                    // vg_api.hip: the plain scan + merge (the filter's pre-pass)
// vg_api.hip; mirror_*: three counter words the merge's workgroup copies on its way (the filter scans' pinned counter mirror)
int vg_launch_merge_one(const uint64_t *dev_cand, int nlists, int k, uint64_t *dev_out_keys, hipStream_t stream,
                        const unsigned long long *mirror_src = nullptr, unsigned long long *mirror_dst = nullptr);
int vg_plain_scan_shape(const vg_corpus *c, int metric, VgShape *out);   // vg_api.hip: launch shape of the plain kernel
int vg_launch_scan_filter(vg_corpus *c, int metric, const uint8_t *dev_query, int k, uint64_t *dev_out_keys, hipStream_t stream,
                          bool ref_emit = false, uint64_t *final_out = nullptr);   // vg_filter.hip; -1: not served
int vg_scan_topk_enqueue_plan(vg_corpus *c, int metric, const void *query, int k, bool ref_emit);   // vg_api.hip: vg_scan_topk_enqueue with the reference-order extras
bool vg_scan_filter_would_serve(const vg_corpus *c, int metric, int k);   // vg_filter.hip: a single scan would take a filter scan right now
bool vg_scan_filter_policy(const vg_corpus *c);      // vg_filter.hip: filter switched on for this corpus and the corpus large enough for the shadow copy to pay
int vg_ensure_filter_counters(vg_corpus *c);         // vg_filter.hip: d_filter_evals[2] + pinned mirror
bool vg_scan_filter_name(vg_corpus *c, int metric, char *out, size_t out_len);   // vg_filter.hip: kernel name when the filter serves the scan
bool vg_batch_keys_are_scan_exact(const vg_corpus *c, int metric, int k);   // vg_batch_api.hip: a batch's floats are the single scan's
long long vg_bf16_shadow_stride(const vg_corpus *c);               // vg_batch_api.hip: row stride of the bf16 shadow copy of an f32 corpus
int vg_ensure_bf16_shadow(vg_corpus *c);                           // vg_batch_api.hip: build / extend it (corpus stream)
int vg_ensure_q8_shadow(vg_corpus *c);                             // vg_filter.hip: the int8 shadow copy + per-row (scale, residual norm)
int vg_multi_queries_per_pass(const vg_corpus *c, int metric);     // vg_multi.hip: queries per pass of the multi-query scan, 0 = none
int vg_launch_scan_multi(vg_corpus *c, int metric, const uint8_t *dev_queries, int k, uint64_t *dev_cand,
                         uint64_t *dev_out_keys, hipStream_t stream);   // vg_multi.hip; -1: no multi-query kernel for this shape
