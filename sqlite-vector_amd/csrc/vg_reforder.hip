// vg_reforder.hip - tie_order = reference: the reference's history-dependent result among EQUAL distances, reproduced
// rowid for rowid (north_star: "bit-exact rowid/top-k ordering for int8/uint8").
//
// One store-mode scan leaves all N distances in HBM (vg_scan_distances_resident, vg_api.hip).  The host replays the
// reference's slot algorithm (vg_refslots.h) over the first P rows, then asks the device for every later row below the
// bound reached so far - vg_below_kernel compacts them, typically a few thousand of millions - and replays those in scan
// order with the exact rule.  Extra cost over the fused top-k scan: 4 bytes written per row, one 4-byte-per-row read
// pass (N = 10M: ~10 us) and two small copies.
#include "vg_internal.h"
#include "vg_refslots.h"


// out[0] = number of rows with dist < bound in [from, n) (may exceed cap), out[1 + i] = (position << 32) | float bits
__global__ __launch_bounds__(256) void vg_below_kernel(const float *dist, long long from, long long n, float bound,
                                                       unsigned long long *out, unsigned cap) {
    const int lane = threadIdx.x & 63;
    const long long q0 = from >> 2, q1 = (n + 3) >> 2;                       // quads of 4 consecutive rows
    for (long long qd = q0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < ((q1 - q0 + 63) / 64 * 64 + q0);
         qd += (long long)gridDim.x * blockDim.x) {
        float v[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        const long long base = qd * 4;
        if (qd < q1) {
            if (base + 3 < n) {
                const float4 f = *reinterpret_cast<const float4 *>(dist + base);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            } else {
                for (int j = 0; j < 4; ++j) if (base + j < n) v[j] = dist[base + j];
            }
        }
        unsigned mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (base + j >= from && base + j < n && v[j] < bound) mine |= 1u << j;
        const int cnt = __popc(mine);
        if (__ballot(mine != 0u) == 0ull) continue;          // nearly every wavefront: nothing below the bound in its 256 rows
        // wave-level slot reservation: exclusive prefix of cnt over the lanes, one atomic per wavefront
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        const int total = __shfl(incl, 63, 64);
        if (total == 0) continue;
        unsigned long long wbase = 0;
        if (lane == 63) wbase = atomicAdd(out, (unsigned long long)total);
        wbase = __shfl(wbase, 63, 64);
        unsigned long long slot = wbase + (unsigned long long)(incl - cnt);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (mine & (1u << j)) {
                if (slot < cap) out[1 + slot] = ((unsigned long long)(base + j) << 32) | (unsigned long long)__float_as_uint(v[j]);
                ++slot;
            }
    }
}

static int ensure_ref_pinned(vg_corpus *c) {
    if (c->h_ref) return VG_OK;
    HIP_TRY(hipHostMalloc(&c->h_ref, VG_REF_PINNED_BYTES));
    c->h_ref_bytes = VG_REF_PINNED_BYTES;
    return VG_OK;
}

extern "C" int vg_resident_distances_fetch(vg_corpus *c, int64_t pos0, int64_t n, float *out_host) {
    if (!c || !out_host) return vg_fail(VG_ERR_INVALID, "vg_resident_distances_fetch: NULL argument");
    if (n <= 0) return VG_OK;
    if (!c->d_dist || pos0 < 0 || pos0 + n > c->dist_valid_rows) return vg_fail(VG_ERR_INVALID, "vg_resident_distances_fetch: no resident distances for rows %lld..%lld", (long long)pos0, (long long)(pos0 + n));
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_ref_pinned(c);
    if (rc != VG_OK) return rc;
    const size_t bytes = (size_t)n * sizeof(float);
    void *dst = bytes <= c->h_ref_bytes ? (void *)c->h_ref : (void *)out_host;        // small copies land in pinned memory
    HIP_TRY(hipMemcpyAsync(dst, c->d_dist + pos0, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (dst != (void *)out_host) memcpy(out_host, dst, bytes);
    return VG_OK;
}

extern "C" int vg_resident_distances_below(vg_corpus *c, int64_t pos0, float bound, uint64_t *out_pairs, int64_t cap, int64_t *out_count) {
    if (!c || !out_pairs || !out_count) return vg_fail(VG_ERR_INVALID, "vg_resident_distances_below: NULL argument");
    *out_count = 0;
    const int64_t n = c->dist_valid_rows;
    if (pos0 >= n) return VG_OK;
    if (!c->d_dist || pos0 < 0) return vg_fail(VG_ERR_INVALID, "vg_resident_distances_below: no resident distances");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_below) HIP_TRY(hipMalloc(&c->d_below, (size_t)(VG_BELOW_CAP + 1) * sizeof(unsigned long long)));
    const unsigned dcap = (unsigned)std::min<int64_t>(cap, VG_BELOW_CAP);
    HIP_TRY(hipMemsetAsync(c->d_below, 0, sizeof(unsigned long long), c->stream));
    const long long quads = ((n + 3) >> 2) - (pos0 >> 2);
    const unsigned blocks = (unsigned)std::max<long long>(1, std::min<long long>((quads + 255) / 256, (long long)c->cu_count * 8));
    hipLaunchKernelGGL(vg_below_kernel, dim3(blocks), dim3(256), 0, c->stream, (const float *)c->d_dist, (long long)pos0, (long long)n, bound,
                       c->d_below, dcap);
    HIP_TRY(hipGetLastError());
    // the count and the first pairs in one copy (into pinned memory); the rest only when there are more
    int rcp = ensure_ref_pinned(c);
    if (rcp != VG_OK) return rcp;
    const size_t first = std::min<size_t>((size_t)dcap, 8191);
    unsigned long long *head = reinterpret_cast<unsigned long long *>(c->h_ref);
    HIP_TRY(hipMemcpyAsync(head, c->d_below, (first + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const unsigned long long count = head[0];
    *out_count = (int64_t)count;
    const size_t have = (size_t)std::min<unsigned long long>(count, dcap);
    memcpy(out_pairs, head + 1, std::min(have, first) * sizeof(uint64_t));
    if (have > first) {
        HIP_TRY(hipMemcpyAsync(out_pairs + first, c->d_below + 1 + first, (have - first) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return VG_OK;
}

namespace {
struct CorpusSrc {
    vg_corpus *c;
    std::vector<uint64_t> &pairs;
    int fetch(int64_t g0, int64_t cnt, float *out) { return vg_resident_distances_fetch(c, g0, cnt, out); }
    int below(int64_t g0, float bound, std::vector<VgRefCand> &out, bool *overflow) {
        if (pairs.size() < (size_t)VG_BELOW_CAP) pairs.resize(VG_BELOW_CAP);      // (once per corpus)
        int64_t count = 0;
        int rc = vg_resident_distances_below(c, g0, bound, pairs.data(), VG_BELOW_CAP, &count);
        if (rc != VG_OK) return rc;
        if (count > VG_BELOW_CAP) { *overflow = true; return VG_OK; }
        for (int64_t i = 0; i < count; ++i) {
            const uint32_t bits = (uint32_t)pairs[(size_t)i];
            float d;
            memcpy(&d, &bits, 4);
            out.push_back(VgRefCand{(int64_t)(pairs[(size_t)i] >> 32), d});
        }
        return VG_OK;
    }
};
}

// the store-mode replay (round 2's form): one scan that writes all N distances, host replay of a prefix, device compaction of the
// later rows below the bound.  Still what serves k >= 64, long rows, small corpora with a tie and candidate-buffer overflows.
static int reference_store_mode_replay(vg_corpus *c, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist, int *out_count) {
    int rc = vg_scan_distances_resident(c, metric, query);
    if (rc != VG_OK) return rc;
    CorpusSrc src{c, c->ref_pairs};
    VgRefSlots slots;
    if ((rc = vg_ref_replay(src, c->n_rows, k, slots)) != VG_OK) return rc;
    vg_collect_timing(c);
    const int cnt = slots.finish();
    for (int i = 0; i < cnt; ++i) {
        out_dist[i] = slots.dist[(size_t)i];
        out_rowids[i] = vg_corpus_rowid_at(c, slots.pos[(size_t)i]);
    }
    *out_count = cnt;
    ++c->ref_stats[3];
    return VG_OK;
}

// one slab of a table scanned slab by slab (vg_slabscan.hip): all of the slab's distances stay on the device, the slots the earlier slabs
// left are offered the slab's rows in scan order (a fresh stream: prefix + candidates; a continued one: the rows below the bound reached)
int vg_ref_replay_slab(vg_corpus *c, int metric, const void *query, int k, VgRefSlots &slots, int64_t gbase, bool fresh) {
    int rc = vg_scan_distances_resident(c, metric, query);
    if (rc != VG_OK) return rc;
    CorpusSrc src{c, c->ref_pairs};
    if (fresh) {
        if ((rc = vg_ref_replay(src, c->n_rows, k, slots)) != VG_OK) return rc;
        if (gbase) for (int i = 0; i < k; ++i) if (slots.pos[(size_t)i] >= 0) slots.pos[(size_t)i] += gbase;
    } else if ((rc = vg_ref_replay_more(src, c->n_rows, k, slots, gbase)) != VG_OK) return rc;
    vg_collect_timing(c);
    ++c->ref_stats[0];
    ++c->ref_stats[3];
    return VG_OK;
}

int vg_ensure_ref_buffers(vg_corpus *c, int64_t prefix_rows) {
    if (c->ref_prefix_cap < prefix_rows) {
        if (c->d_ref_prefix) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->d_ref_prefix); c->d_ref_prefix = nullptr; c->ref_prefix_cap = 0; }
        HIP_TRY(hipMalloc(&c->d_ref_prefix, (size_t)prefix_rows * sizeof(float)));
        c->ref_prefix_cap = prefix_rows;
    }
    if (!c->d_below) HIP_TRY(hipMalloc(&c->d_below, (size_t)(VG_BELOW_CAP + 1) * sizeof(unsigned long long)));
    return ensure_ref_pinned(c);
}

// What the last emitting launch left on the device comes to the host in two steps, so that a caller with several shards can put all
// the copies in flight before it waits for any: the prefix pass' distances (rows 0 .. P-1 of this corpus, all of them) and the
// candidate stream (every later row that can enter the reference's slots; a superset, in any order).
int vg_ref_emitted_enqueue(vg_corpus *c) {
    const int64_t P = c->ref_prefix_rows;
    if (P <= 0 || !c->d_ref_prefix || !c->d_below || !c->h_ref) return vg_fail(VG_ERR_INVALID, "vg_ref_emitted_enqueue: the last launch did not emit");
    HIP_TRY(hipSetDevice(c->device));
    float *prefix = reinterpret_cast<float *>(c->h_ref);
    unsigned long long *head = reinterpret_cast<unsigned long long *>(c->h_ref + (size_t)VG_REF_PREFIX_MAX * 4);
    HIP_TRY(hipMemcpyAsync(prefix, c->d_ref_prefix, (size_t)P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(head, c->d_below, (VG_REF_FIRST_PAIRS + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    return VG_OK;
}
// waits for those copies; *overflow: the stream held more pairs than the device buffer does (the caller takes the store-mode replay).
// The pointers stay valid until the next reference-order call on this corpus.
int vg_ref_emitted_wait(vg_corpus *c, const float **prefix, int64_t *prefix_rows, unsigned long long **pairs, unsigned long long *count,
                        bool *overflow) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long *head = reinterpret_cast<unsigned long long *>(c->h_ref + (size_t)VG_REF_PREFIX_MAX * 4);
    const unsigned long long n = head[0];
    *overflow = n > (unsigned long long)VG_BELOW_CAP;
    *prefix = reinterpret_cast<const float *>(c->h_ref);
    *prefix_rows = c->ref_prefix_rows;
    *pairs = head + 1;
    *count = *overflow ? 0 : n;
    if (!*overflow && n > VG_REF_FIRST_PAIRS) {
        HIP_TRY(hipMemcpyAsync(head + 1 + VG_REF_FIRST_PAIRS, c->d_below + 1 + VG_REF_FIRST_PAIRS, (size_t)(n - VG_REF_FIRST_PAIRS) * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return VG_OK;
}

// a run of consecutive rows (positions g0 ..) offered to the slots.  Nearly every row is no candidate once the slots are warm: look at
// 16 rows at a time (a min the compiler vectorises)
void vg_ref_offer_run(VgRefSlots &slots, const float *d, int64_t n, int64_t g0) {
    int64_t i = 0;
    for (; i + 16 <= n; i += 16) {
        float m = d[i];
        for (int j = 1; j < 16; ++j) m = d[i + j] < m ? d[i + j] : m;               // (NaN never wins: it cannot enter either)
        if (!((double)m < slots.cur_max) && !(d[i] != d[i])) continue;              // (a leading NaN would hide the rest: take the slow way)
        for (int j = 0; j < 16; ++j) slots.offer(d[i + j], g0 + i + j);
    }
    for (; i < n; ++i) slots.offer(d[i], g0 + i);
}

// The reference's slots replayed over them.  *overflow as above.
static int replay_emitted(vg_corpus *c, int k, VgRefSlots &slots, bool *overflow) {
    int rc = vg_ref_emitted_enqueue(c);
    if (rc != VG_OK) return rc;
    const float *prefix = nullptr;
    int64_t P = 0;
    unsigned long long *pairs = nullptr, count = 0;
    if ((rc = vg_ref_emitted_wait(c, &prefix, &P, &pairs, &count, overflow)) != VG_OK) return rc;
    if (*overflow) return VG_OK;
    slots.init(k);
    vg_ref_offer_run(slots, prefix, P, 0);
    // the rows behind it, in scan order (position is the high word of a pair)
    std::sort(pairs, pairs + count);
    for (unsigned long long j = 0; j < count; ++j) {
        const int64_t pos = (int64_t)(pairs[j] >> 32);
        if (pos < P) continue;                              // (the main pass covers the prefix rows again)
        const uint32_t bits = (uint32_t)pairs[j];
        float d;
        memcpy(&d, &bits, 4);
        slots.offer(d, pos);
    }
    return VG_OK;
}

// tie_order = reference.  The reference's result differs from (distance, position) order only when equal distances meet among the
// k+1 best: with k+1 pairwise distinct distances its k slots end up holding exactly the k smallest, and its exchange sort of
// distinct values is the ascending order.  So: the ordinary top-k scan with ONE MORE list slot; no tie among those k+1 -> done, the
// same cost as tie_order = position.  A tie -> the slots are replayed on the host over the rows that can enter them at all
// (replay_emitted), which the scan left behind at no cost: no second pass over the corpus, no N-distance store.
extern "C" int vg_scan_topk_reference(vg_corpus *c, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist,
                                      int *out_count) {
    if (!c || !query || !out_count) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_reference: NULL argument");
    *out_count = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_rowids || !out_dist) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_reference: NULL output");
    if (vg_metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    ++c->ref_stats[0];
    VgShape shape;
    vg_plain_scan_shape(c, metric, &shape);
    const bool can_emit = !shape.long_rows;                 // (rows too long for a register-resident shape have no emitting kernel)
    // k = 64: the lists have no 65th slot to look for a tie with - the scan runs emitting with k slots and the slots are ALWAYS replayed
    // (the candidate stream of k-slot lists is still a superset of the rows that can enter k slots)
    const bool always = (k == VG_WAVE_HOST);
    if (k > VG_WAVE_HOST || (always && !can_emit) || vg_sw(SW_VG_REF_STORE_MODE, 0))
        return reference_store_mode_replay(c, metric, query, k, out_rowids, out_dist, out_count);
    const int k1 = always ? k : k + 1;
    // scans through a filter kernel have a pre-pass anyway (emitting is free); plain-kernel scans pay for one only while ties are around
    bool emit = can_emit && (always || c->ref_hot > 0 || vg_scan_filter_would_serve(c, metric, k1) || vg_sw(SW_VG_REF_ALWAYS_EMIT, 0));
    for (int attempt = 0; attempt < 2; ++attempt) {
        uint64_t keys[64];
        int rc = vg_scan_topk_enqueue_plan(c, metric, query, k1, emit);
        if (rc == VG_OK) rc = vg_scan_topk_collect(c, keys);
        if (rc != VG_OK) return rc;
        int cnt = 0;
        while (cnt < k1 && keys[cnt] != VG_KEY_EMPTY) ++cnt;
        bool tie = always;
        for (int i = 1; i < cnt; ++i) tie |= (keys[i] >> 32) == (keys[i - 1] >> 32);
        if (!tie) {
            const int take = std::min(cnt, k);
            for (int i = 0; i < take; ++i) {
                out_dist[i] = (double)vg_key_distance(keys[i]);
                out_rowids[i] = vg_corpus_rowid_at(c, (int64_t)vg_key_position(keys[i]));
            }
            *out_count = take;
            if (c->ref_hot > 0) --c->ref_hot;
            return VG_OK;
        }
        if (attempt == 0) ++c->ref_stats[1];
        if (!always) c->ref_hot = 16;   // (break-even: a plain-kernel scan pays ~85 us for prefix pass + emitting kernel, a tie without them a
                                        // second ~1.2 ms scan - emitting pays from one tie in ~14 queries on)
        if (emit && c->ref_prefix_rows > 0) {
            VgRefSlots slots;
            bool overflow = false;
            rc = replay_emitted(c, k, slots, &overflow);
            if (rc != VG_OK) return rc;
            if (overflow) break;                            // too many candidates (heavy ties / descending distances): store mode
            const int n = slots.finish();
            for (int i = 0; i < n; ++i) {
                out_dist[i] = slots.dist[(size_t)i];
                out_rowids[i] = vg_corpus_rowid_at(c, slots.pos[(size_t)i]);
            }
            *out_count = n;
            ++c->ref_stats[2];
            return VG_OK;
        }
        if (emit || !can_emit) break;                       // this corpus cannot emit (long rows): store mode
        emit = true;                                        // a tie and nothing emitted: scan again, emitting
    }
    return reference_store_mode_replay(c, metric, query, k, out_rowids, out_dist, out_count);
}

extern "C" int vg_corpus_tie_stats(const vg_corpus *c, unsigned long long *out4) {
    if (!c || !out4) return vg_fail(VG_ERR_INVALID, "vg_corpus_tie_stats: NULL argument");
    for (int i = 0; i < 4; ++i) out4[i] = c->ref_stats[i];
    return VG_OK;
}

// The same replay over distances the caller already holds on the host (e.g. a *_stream result): out_pos are scan
// positions.  below_cap (> 0) bounds the candidate set the way the device buffer does - what the CPU tests use to drive
// the "too many candidates" path of the driver without a GPU.
namespace {
struct HostSrc {
    const float *d; int64_t n; int64_t cap;
    int fetch(int64_t g0, int64_t cnt, float *out) { memcpy(out, d + g0, (size_t)cnt * sizeof(float)); return 0; }
    int below(int64_t g0, float bound, std::vector<VgRefCand> &out, bool *overflow) {
        for (int64_t i = n - 1; i >= g0; --i)                 // (any order: the driver sorts by position)
            if (d[i] < bound) {
                if ((int64_t)out.size() >= cap) { out.clear(); *overflow = true; return 0; }
                out.push_back(VgRefCand{i, d[i]});
            }
        return 0;
    }
};
}
extern "C" int vg_reference_topk_replay(const float *dist, int64_t n, int k, int64_t below_cap, int64_t *out_pos, double *out_dist) {
    if (!dist || n < 0 || k < 0 || !out_pos || !out_dist) { vg_fail(VG_ERR_INVALID, "vg_reference_topk_replay: bad argument"); return -1; }
    HostSrc src{dist, n, below_cap > 0 ? below_cap : VG_BELOW_CAP};
    VgRefSlots slots;
    if (vg_ref_replay(src, n, k, slots) != 0) return -1;
    const int cnt = slots.finish();
    for (int i = 0; i < cnt; ++i) { out_pos[i] = slots.pos[(size_t)i]; out_dist[i] = slots.dist[(size_t)i]; }
    return cnt;
}

// the same stream handed over SLAB BY SLAB (what vg_slabscan.hip does with a table that does not fit the device): the first slab is replayed
// like a corpus, every later one continues the slots it inherits (vg_ref_replay_more) - a host-only twin of vg_ref_replay_slab for the CPU tests
extern "C" int vg_reference_topk_replay_slabs(const float *dist, int64_t n, int k, int64_t slab_rows, int64_t below_cap, int64_t *out_pos, double *out_dist) {
    if (!dist || n < 0 || k < 0 || slab_rows < 1 || !out_pos || !out_dist) { vg_fail(VG_ERR_INVALID, "vg_reference_topk_replay_slabs: bad argument"); return -1; }
    VgRefSlots slots;
    slots.init(k);
    bool fresh = true;
    for (int64_t base = 0; base < n; base += slab_rows) {
        const int64_t cnt = std::min<int64_t>(slab_rows, n - base);
        HostSrc src{dist + base, cnt, below_cap > 0 ? below_cap : VG_BELOW_CAP};
        if (fresh) {
            if (vg_ref_replay(src, cnt, k, slots) != 0) return -1;
            if (base) for (int i = 0; i < k; ++i) if (slots.pos[(size_t)i] >= 0) slots.pos[(size_t)i] += base;
            fresh = false;
        } else if (vg_ref_replay_more(src, cnt, k, slots, base) != 0) return -1;
    }
    const int got = slots.finish();
    for (int i = 0; i < got; ++i) { out_pos[i] = slots.pos[(size_t)i]; out_dist[i] = slots.dist[(size_t)i]; }
    return got;
}

extern "C" int vg_corpus_set_tie_order(vg_corpus *c, int mode) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    if (mode != VG_TIE_POSITION && mode != VG_TIE_REFERENCE) return vg_fail(VG_ERR_INVALID, "unknown tie order %d", mode);
    c->tie_order = mode;
    return VG_OK;
}
extern "C" int vg_corpus_tie_order(const vg_corpus *c) { return c ? c->tie_order : 0; }
