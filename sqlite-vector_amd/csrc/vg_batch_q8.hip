// vg_batch_q8.hip - batched queries over an f32 corpus (rows up to 512 floats) with the INTEGER matrix cores as the filter:
// Q x C^T on v_mfma_i32_32x32x32_i8 over the corpus' int8 shadow copy (the one the single-query filter scan streams, vg_filter.hip:
// per row x = sx * xi + ex, integers xi in [-127, 127], sx = max|x| / 127, ||ex|| stored rounded up), the single-query kernel's own
// f32 arithmetic for the pairs that pass (vg_batch_hx_kernel, vg_batch_h_defs.h).  The reference has no batched entry point: the
// oracle of a batch is Q independent vFullScanRun calls (sqlite-vector.c:2071-2113) over distance-avx2.c:67-162.
//
// Why (round 5): the bf16 filter (vg_batch_h.hip) is bound by what surrounds its MFMAs - one 1 KiB B-operand read from LDS, one wait
// and, per tile, the LDS-DMA issues and the barrier, all per 32 queries x 32 rows x 16 elements of matrix work.  int8 halves the A
// operand (32 queries x 384 elements = 48 registers instead of 96), so a wavefront keeps TWO sets of 32 queries stationary and every
// B read, every wait, every DMA piece and every barrier serves twice the matrix work at twice the K per instruction; the shadow copy
// streamed is a quarter of the corpus (3.84 GB at 10M x 384) instead of half.
//
// The bound (vg_scan_filter.h, Q8 = true; no rounding argument involved - whatever integers the quantizers picked, the residuals are
// what is left):  q = sq qi + eq,  x = sx xi + ex,
//     q.x = sq sx (qi.xi) + sq (qi.ex) + eq.x        |q.x - sq sx (qi.xi)| <= sq ||qi|| ||ex|| + ||eq|| ||x||        (Cauchy-Schwarz)
// with qi.xi an exact int32 from the matrix core.  A pair passes when the lower bound of its distance can beat the query's k-th best
// so far; written per pair as ONE sum that must not be negative (t = (float)(qi.xi) * sx):
//     dot      t sq + A ||ex|| + B ||x|| + gate(thr)                     >= 0      A = sq ||qi||,  B = ||eq|| + rel |q| (+ roundings)
//     L2       t sq + A ||ex|| + B ||x|| - (1 - rel) |x|^2 / 2 + (gate(thr^2) - (1 - rel) |q|^2) / 2 >= 0
//     cosine   t sq + A ||ex|| + B ||x|| - G(thr) |q| ||x||              >= 0      G = 1 - 4e-6 - gate(thr)
// The tile boundary works in the query's own integer units (everything divided by sq): the lane's LARGEST accumulator of a query set is
// compared, as an integer, with the set's loosest gate; the batch is SORTED by the norm of the int8 images so that the 32 queries of a
// set share nearly the same gate (see vg_q8_query_prep_kernel).  Only accumulators that pass get the query's own four coefficients
// (LDS), and the pairs that pass those go to the wavefront's two regions of the pair buffer (one per query set) in scan order.
// Thresholds are fixed per launch and refreshed between stages over growing row ranges (the split form of vg_batch_h.hip); the first
// stage - two tiles - lets every pair through and hands the next one exact lists.
// Workgroup = FOUR wavefronts (one per SIMD), TWO workgroups per CU: the tile barrier couples four wavefronts, and while one workgroup's
// wavefront waits (barrier, a tile with candidates) the other workgroup's wavefront on the same SIMD keeps the matrix pipe busy.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "vg_accum.h"
#include "vg_batch_common.h"
#include "vg_batch_h_defs.h"

typedef int vgq_i32x16 __attribute__((ext_vector_type(16)));

#define VGQ_WAVES 4                     // one wavefront per SIMD; two workgroups per CU
#define VGQ_QS 2                        // query sets of 32 per wavefront
#define VGQ_QPW (32 * VGQ_QS)
#define VGQ_QPB (VGQ_WAVES * VGQ_QPW)   // 256 queries per workgroup
// LONG rows (513 .. 1536 elements; round 5): a row's whole A operand for two query sets does not fit the register file (2 x 48 k-steps x 4 =
// 384 VGPRs at 1536) - ONE set per wavefront (192), EIGHT wavefronts per workgroup (still 256 queries per pass over the copy), one workgroup
// per CU, and the ring moves SUB-tiles: 32 rows x NTB k-steps = one K-part of a tile (contiguous in the tile-major copy); the accumulators
// are carried over a tile's KS parts and the tile boundary runs behind the last one.
#ifndef VGQL_WAVES
#define VGQL_WAVES 8
#endif
#ifndef VGQL_QS
#define VGQL_QS 1
#endif
#define VGQL_RING 6
#define VGQ_TILE 32
#define VGQ_MAX_K 64                    // (round 6: one list slot per lane of the exact-evaluation wavefront - the single scans' own limit)
#define VGQ_BPIPE 4
#define VGQ_RING_OF(NTB) ((NTB) <= 8 ? 6 : (NTB) <= 12 ? 4 : 3)      // tile buffers: two workgroups' rings + statistics + queues fit 160 KB
#ifndef VGQW_RING
#define VGQW_RING 6                     // tile buffers of the wide form (eight wavefronts x two query sets, one workgroup per CU)
#endif
#define VGQW_RING_OF(NTB) ((VGQW_RING) * (NTB) <= 112 ? (VGQW_RING) : 6)
#define VGQ_QCAP 16                     // candidate lanes a wavefront collects before it looks at their accumulators (160 bytes each)   (= 64 lanes / 4 lanes per entry: a queue run looks at exactly that many)
#ifndef VGQ_NARROW_PARTS
#define VGQ_NARROW_PARTS 512            // partitions of the 128-slot form (one query group: 512 workgroups = two per CU)
#endif
#ifndef VGQ_NARROW
#define VGQ_NARROW 1                    // batches of up to 128 queries over short rows: the 128-slot form (0: the 256-slot form, as before)
#endif
#ifndef VGQ_DRAIN_MIN
#define VGQ_DRAIN_MIN 4                 // parked lanes a wavefront waits for before a tile retires two of them
#endif
#ifndef VGQ_SWP_DRAIN
#define VGQ_SWP_DRAIN 1                 // the pipelined form retires two parked candidate lanes per tile beside its MFMAs (see the kernel)
#endif
#ifndef VGQ_SWP
#define VGQ_SWP 1                       // 1: the tile boundary's first test software-pipelined under the next tile's MFMAs (see the kernel)
#endif
#define VGQ_STAT_SLOTS 8                // ring of row-statistics groups (two tiles = 1 KiB each)
#define VGQ_STAGE0_TILES 2              // the first stage: every pair passes (32 queries x 32 rows per region and tile <= the pair capacity)
#ifndef VGQ_TPB
#define VGQ_TPB 3                       // short rows: ring trips (tiles) per workgroup barrier where the ring has six buffers (see the tile loop)
#endif
#ifndef VGQ_PRE_TILES
#define VGQ_PRE_TILES 512              // the bound-only pre-pass: tiles it covers (at most 1/8 of the corpus; <= 2048: vg_q8_pre_select_kernel keeps 32 tile minima per lane)
#endif
#ifndef VGQ_HX_GROUP
#define VGQ_HX_GROUP 4                  // partitions one exact-evaluation block walks at most (BatchArgsH.part_group)
#endif
#ifndef VGQ_HX_GROUP_FIRST
#define VGQ_HX_GROUP_FIRST 0
#endif
#ifndef VGQ_FIRST_MULT
#define VGQ_FIRST_MULT 2                // the first real stage ends at VGQ_FIRST_MULT x the pre-pass' tiles (long rows: 1 x - their bound is twice as wide
#endif                                  // against the spread of the scores, and an exact evaluation reads up to 6 KB); measured: profiles/r10_q8_schedule_sweep.txt
#ifndef VGQ_ABLATE
#define VGQ_ABLATE 0                    // measurement builds (wrong results): 1 = candidates dropped; 2 = no gate; 3 = + no LDS-DMA; 4 = + no barrier
#endif
#ifndef VGQ_STATS
#define VGQ_STATS 0                     // measurement builds: counters of the tile boundary (wave-tiles | that went on to single accumulators | candidates | pairs)
#endif
#ifndef VGQ_TIMING
#define VGQ_TIMING 0                    // measurement builds: shader-clock ticks per wavefront class (tools/r6_q8_timing.py)
#endif
#if VGQ_TIMING
// [0..7] wavefronts 0-3, [8..15] wavefronts 4-7 of the eight-wavefront forms: whole tile loop | k loops | boundaries | trip-end wait + barrier | wave-tiles | boundaries that queued a candidate | queue runs | -
__device__ unsigned long long vgq_ticks[16];
extern "C" int vg_batch_q8_timing(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(vgq_ticks), sizeof(vgq_ticks)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vgq_ticks), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define VGQ_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define VGQ_TICK(var)
#endif
#ifndef VGQ_TRACE
#define VGQ_TRACE 0                     // measurement builds: per-tile timestamps of ONE workgroup's eight wavefronts (tools/r6_q8_tile_trace.py)
#endif
#if VGQ_TRACE
#define VGQ_TRACE_TILES 96
__device__ unsigned long long vgq_trace[8 * VGQ_TRACE_TILES * 4];     // [wave][tile][k loop begins | k loop ends | boundary ends | past the wait / barrier]
extern "C" int vg_batch_q8_trace(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(vgq_trace), sizeof(vgq_trace)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[8 * VGQ_TRACE_TILES * 4]; if (hipMemcpyToSymbol(HIP_SYMBOL(vgq_trace), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
#if VGQ_STATS
__device__ unsigned long long vgq_stats[32];        // [6 stage classes by tiles per partition][wave-tiles, tiles with a candidate lane, candidate lanes, registers past the integer test, pairs]
extern "C" int vg_batch_q8_stats(unsigned long long *out8, int reset) {        // (out8: 32 values)
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(vgq_stats), sizeof(vgq_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vgq_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

struct BatchArgsQ8 {
    const uint8_t *rows;      // the TILE-MAJOR int8 shadow copy: tile t = rows 32t .. 32t+31 = 32 * stride contiguous bytes, chunk column c
                              // of the 32 rows at c * 512 + row * 16 (vg_tile_major_kernel over the row-major shadow copy)
    const float4 *rstat;      // per row: (sx, ||ex|| rounded up, ||x||, 1 / sx); a row that is never judged: (1e-30, 3e38, -1, 1e30); readable for two tiles past the last one
    const uint8_t *qcodes;    // nq_pad x stride int8 query images (vg_q8_query_prep_kernel), zero padded, in SORTED order (see there)
    const float4 *qstat;      // per query two float4: (sq, sq ||qi|| up, ||eq|| up, |q|), (|q|^2, judged ? 1 : 0, 0, 0)
    long long n_rows, stride; // stride: bytes per int8 row (multiple of 16)
    int nq_pad, npart, k, mode, root;
    float rel;                // (D + 64) 2^-22: what the cached norms and the exact evaluation themselves may be off by (vg_filter.hip)
    int tiles_per_part;
    long long tile_begin, tile_end;
    int part_base, npart_total;
    const uint64_t *init_keys; // NULL: every pair of a judged query passes (the first stage)
    uint64_t *pairs;          // region = ((g * npart_total + part_base + part) * 8 + wave * 2 + set); a pair = (query in the set) << 32 | row
    uint32_t *pair_counts;    // [region] pairs written; [flag_index] = overflow flag
    int pair_cap, flag_index;
    float *premin;            // PRE kernels: [tile - tile_begin][nq_pad] - per query slot the SMALLEST upper bound of the distance over the tile's 32 rows
};

__device__ __forceinline__ int vgq_q8(float v, float inv) {                 // (vgf_q8 of vg_scan_filter.h)
    const float t = rintf(v * inv);
    return (int)fminf(fmaxf(t, -127.0f), 127.0f);
}
#define VGQ_JUDGE_LO 1.0e-10f           // queries / rows whose magnitude lies outside [LO, HI] are not judged (the gate divides by the query's scale)
#define VGQ_JUDGE_HI 1.0e10f

// ---- the queries' int8 images and statistics, one wavefront per query SLOT.  Slot p holds query perm[p] (perm == NULL: p): the batch is
// SORTED by what a query's gate on the integer score is proportional to, so that the 32 queries of a set share nearly the same gate and one
// integer comparison of the lane's largest accumulator against the set's loosest gate is tight (unsorted, with the wavefront's largest
// coefficients: 64 % of the tiles went on to single pairs; docs/ROUND5_NOTEBOOK.md §2).  dot / cosine: the norm of the int8 image under the query's OWN
// scale sq = max|q| / 127 (threshold / sq ~ z |q| / sq).  L2: the gate holds |x|^2 / sq and (thr^2 - |q|^2) / sq - two query-dependent
// factors - so L2 batches share ONE scale (vg_q8_rank_kernel; a query whose own scale is larger, or 8 x smaller, is not judged) and sort by |q|.
// keys_out != NULL: only the sort key and the own scale of every query (original order) are written.
// (qtype: the element type of the queries = the corpus' own: 0 f16, 1 bf16, 2 f32 - the int8 image is taken of the widened value)
__global__ __launch_bounds__(256) void vg_q8_query_prep_kernel(const uint8_t *xq, long long xstride, int dim, int nq_real, int nq_pad, int mode,
                                                               const int *perm, const float *common_scale, const float *all_scales, float *keys_out, float *scales_out,
                                                               uint8_t *xq_sorted, uint8_t *codes, long long qstride, float4 *qstat, int qtype) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (p >= nq_pad) return;
    const int q = perm ? perm[p] : p;
    const uint8_t *srcb = xq + (long long)q * xstride;
    auto elem = [&](int e) -> float {
        if (qtype == 2) return reinterpret_cast<const float *>(srcb)[e];
        const uint32_t b = reinterpret_cast<const uint16_t *>(srcb)[e];
        return qtype == 0 ? vg_h2f(b) : vg_b2f(b);
    };
    float mx = 0.0f;
    uint32_t bad = 0;
    for (int e = lane; e < dim; e += 64) { const float f = elem(e); mx = fmaxf(mx, fabsf(f)); bad |= !(fabsf(f) <= 3.0e38f) ? 1u : 0u; }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
    bool ok = q < nq_real && __ballot(bad != 0) == 0ull && mx >= VGQ_JUDGE_LO && mx <= VGQ_JUDGE_HI;
    float sq = ok ? mx / 127.0f : 1.0f;
    if (common_scale && mode == VGH_L2) {                                 // (phase 2 of an L2 batch)
        // the shared scale = the largest own scale that is at most 4 x the scale of the batch's MEDIAN query (*common_scale, vg_q8_rank_kernel):
        // 16 loads per lane + one reduction per wavefront - cheaper than a launch of its own
        const float ref4 = 4.0f * *common_scale;
        float cs = 0.0f;
        for (int j = lane; j < nq_pad; j += 64) { const float sv = all_scales[j]; if (sv <= ref4) cs = fmaxf(cs, sv); }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) cs = fmaxf(cs, __shfl_xor(cs, s));
        if (ok) {
        ok = sq <= cs && sq * 8.0f >= cs;                                 // (its elements fit the shared grid, and use at least four bits of it)
        sq = ok ? cs : 1.0f;
        }
    }
    const float inv = 1.0f / sq;
    uint32_t i2 = 0;
    float e2s = 0.0f, q2s = 0.0f;
    for (int e = lane; e < (int)qstride; e += 64) {
        const float f = (ok && e < dim) ? elem(e) : 0.0f;
        const int qi = vgq_q8(f, inv);
        const float r = fmaf(-sq, (float)qi, f);                          // one rounding
        i2 += (uint32_t)(qi * qi);
        e2s = fmaf(r, r, e2s);
        q2s = fmaf(f, f, q2s);
        if (codes) codes[(long long)p * qstride + e] = (uint8_t)(qi & 255);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { i2 += __shfl_xor(i2, s); e2s += __shfl_xor(e2s, s); q2s += __shfl_xor(q2s, s); }
    const bool judged = ok && q2s >= VGQ_JUDGE_LO * VGQ_JUDGE_LO && q2s <= VGQ_JUDGE_HI * VGQ_JUDGE_HI;
    if (keys_out) {
        if (lane == 0) { keys_out[p] = judged ? (mode == VGH_L2 ? sqrtf(q2s) : sqrtf((float)i2)) : INFINITY; scales_out[p] = judged ? sq : 0.0f; }
        return;
    }
    if (xq_sorted)
        for (int c = lane; c < (int)(xstride / 16); c += 64)
            reinterpret_cast<uint4 *>(xq_sorted + (long long)p * xstride)[c] = reinterpret_cast<const uint4 *>(xq + (long long)q * xstride)[c];
    if (lane == 0) {
        const float sqi = sq * sqrtf((float)i2) * (1.0f + 1.0e-5f);
        const float eqn = sqrtf(e2s) * (1.0f + 1.0e-4f);                  // (f32 sum of D squares, each off by 2^-23 of itself at most)
        qstat[2 * p] = make_float4(sq, sqi, eqn, sqrtf(q2s));
        qstat[2 * p + 1] = make_float4(q2s, judged ? 1.0f : 0.0f, 0.0f, 0.0f);
    }
}
// perm[rank of query i by (key, i)] = i, and the scale of the batch's MEDIAN query (by key) - what an L2 batch's shared scale is taken
// from (vg_q8_query_prep_kernel, phase 2: one query with a single huge element must not take the int8 grid away from all the others; it is
// answered by a single scan instead).  Round 5 ran this in ONE workgroup: 46 us of a 5 ms batch.
__global__ __launch_bounds__(256) void vg_q8_rank_kernel(const float *keys, const float *scales, int nq_pad, int *perm, float *median_scale) {
    // one WAVEFRONT per query: lane l compares keys l, l + 64, ... (the 4 .. 16 KB of keys stay in the L2) - 1 024 wavefronts of 16 loads each
    // instead of 1 024 threads of 1 024 LDS reads each (8 us against 32)
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= nq_pad) return;
    const float ki = keys[i];
    int rank = 0, judged = 0;
    for (int j = lane; j < nq_pad; j += 64) {
        const float kj = keys[j];
        rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
        judged += (kj < INFINITY) ? 1 : 0;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { rank += __shfl_xor(rank, s); judged += __shfl_xor(judged, s); }
    if (lane == 0) {
        perm[rank] = i;
        if (judged > 0 && rank == judged / 2) *median_scale = scales[i];     // (the judged queries hold the first `judged` ranks; one writer)
        if (judged == 0 && i == 0) *median_scale = 0.0f;
    }
}

// ---- per-row statistics in the layout the filter's LDS-DMA moves (16 bytes per row)
// (nn_squared: f16 / bf16 corpora cache (float) sum x^2 per row, f32 corpora ||x|| itself)
__global__ __launch_bounds__(256) void vg_q8_rstat_kernel(const float2 *q8stat, const float *xnorm, long long row0, long long n, float4 *out, int nn_squared) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 s = q8stat[row0 + i];
        const float nrm = nn_squared ? sqrtf(xnorm[row0 + i]) : xnorm[row0 + i];
        const bool zero = (nrm == 0.0f) && (s.x == 0.0f) && (s.y == 0.0f);                              // every element is +-0
        const bool judged = zero || ((nrm >= VGQ_JUDGE_LO && nrm <= VGQ_JUDGE_HI) && (s.x > 0.0f && s.x <= 3.0e38f));   // (sx = NaN: Inf / NaN elements)
        // a row that is never judged reads as "every gate open" without a test of its own: a huge residual norm drives the integer
        // threshold to -2^31 (the conversion saturates); ||x|| = -1 marks it for the per-pair test
        // (.w = 1 / sx, correctly rounded, made HERE once per row: the filter's tile boundary divided by sx per wave-tile - a ten-instruction
        //  sequence out of ~85; round 6)
        out[row0 + i] = judged ? (zero ? make_float4(0.0f, 0.0f, 0.0f, __frcp_rn(0.0f)) : make_float4(s.x, s.y, nrm, __frcp_rn(s.x)))
                               : make_float4(1.0e-30f, 3.0e38f, -1.0f, __frcp_rn(1.0e-30f));
    }
}

// ---- the start thresholds out of the pre-pass: per query slot the k-th smallest of its tile minima (premin[tile][slot], n_tiles <= 2048),
// one wavefront per slot: 32 values per lane, k rounds of "take the smallest out".  Written as the slot's k-th KEY (what the filter and the
// exact-evaluation kernels read a start threshold from), one ulp-ish above the bound: their tests are strict.  Fewer than k witnesses: no
// threshold (every gate open).
__global__ __launch_bounds__(256) void vg_q8_pre_select_kernel(const float *premin, int n_tiles, int nq_pad, int k, uint64_t *out_keys) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (p >= nq_pad) return;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { const int t = lane + 64 * i; v[i] = t < n_tiles ? premin[(long long)t * nq_pad + p] : INFINITY; }
    float kth = INFINITY;
    for (int round = 0; round < k; ++round) {
        float m = v[0];
        int mi = 0;
#pragma unroll
        for (int i = 1; i < 32; ++i) if (v[i] < m) { m = v[i]; mi = i; }
        float w = m;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) w = fminf(w, __shfl_xor(w, sft));
        kth = w;
        if (!(w < INFINITY)) break;                                       // (fewer than k finite minima)
        const unsigned long long holders = __ballot(m == w);
        if (lane == __ffsll((long long)holders) - 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) if (i == mi) v[i] = INFINITY;
        }
    }
    for (int j = lane; j < 64; j += 64) {
        uint64_t key = VG_EMPTY_KEY;
        if (j == k - 1 && kth < INFINITY) key = vg_make_key(kth + 1.0e-6f * fabsf(kth) + 1.0e-30f, 0xFFFFFFFFu);
        out_keys[(long long)p * 64 + j] = key;
    }
}

template <int OFF>
__device__ __forceinline__ void vgq_lds_read128(vgh_i32x4 &dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void vgq_wait_lds(vgh_i32x4 &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}

// NTB = 32-byte k-steps per int8 row (rows up to NTB * 32 elements)
// WAVES x QS x 32 = 256 queries per workgroup; KS = K-parts per tile (1: the whole row is one ring buffer)
// PRE = the bound-only pre-pass (round 6): no gate, no pairs - per (query, tile) the smallest UPPER bound of the distance (see pre_boundary)
template <int NTB, int MODE, int WAVES = VGQ_WAVES, int QS = VGQ_QS, int KS = 1, bool PRE = false>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 4 && KS == 1 ? 2 : 1)) void vg_batch_q8_kernel(BatchArgsQ8 a) {
    constexpr int THREADS = 64 * WAVES, QPW = 32 * QS, QENT = 16 * QS + 8, QPB = WAVES * QPW;
    constexpr int NB = (WAVES == 4 && KS == 1) ? VGQ_RING_OF(NTB) : (KS == 1 && QS == 2 ? VGQW_RING_OF(NTB) : VGQL_RING);   // ring buffers in LDS; one workgroup per CU: six
    constexpr bool COS = (MODE == VGH_COS), L2M = (MODE == VGH_L2);
    constexpr int TILE_BYTES = NTB * 2 * 512;
    // TRIPS PER BARRIER (round 6).  Measured with a cycle counter per wavefront (profiles/r10_q8_cycles_per_wave_tile.txt): the SIMD's arbiter
    // favours the older of its two wavefronts, so waves 0-3 finish a tile's MFMAs after ~1 000 cycles and waves 4-7 after ~1 650; then the
    // younger half runs its boundary and everybody meets at the barrier - ~1 300 of a tile's ~3 000 cycles with no MFMA in the pipe, because a
    // wavefront that is done with tile t may not start tile t + 1.  The ring is six tiles deep, so the barrier can wait: M trips form a GROUP,
    // the workgroup meets once per group, and inside a group a wavefront goes from tile to tile on its own (the older half's next k loop runs
    // under the younger half's boundary).  Group g reads buffers that landed before its first tile; its tiles issue the DMA of a later group into the
    // buffers of group g - 1, which every wavefront has left (the barrier in between).  M = 1 is the barrier per trip of round 5.
    // (K-parts: a group is one tile's KS trips - the boundary sits behind the last of them anyway)
    constexpr int M = KS > 1 ? (VGQ_TPB > 1 && NB % KS == 0 && NB >= 2 * KS ? KS : 1)
                             : (NB % VGQ_TPB == 0 && NB >= 2 * VGQ_TPB ? VGQ_TPB : (NB % 2 == 0 && NB >= 4 ? 2 : 1));
    constexpr int LOOK = NB / M - 1;                                 // groups beyond the current one that are resident or on their way
    static_assert(LOOK >= 1 && NB % M == 0, "ring of whole groups");
    // (WHO ISSUES THE LDS-DMA: every wavefront its share, in front of its k loop.  Measured and removed in round 6: the four favoured wavefronts
    //  issuing a whole group behind their last boundary of a group - two groups ahead, nine ring buffers - 5.20 against 5.16 ms, and again beside
    //  the pipelined first test 4.47 against 4.37; the four favoured wavefronts issuing three pieces each at the usual place: 4.48-4.50 both ways)
    constexpr int NISSUE = WAVES;                                    // wavefronts that issue tile pieces
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *tile0 = smem;
    float4 *rstat_lds = reinterpret_cast<float4 *>(smem + NB * TILE_BYTES);              // [VGQ_STAT_SLOTS][2 tiles][32 rows]
    float4 *kq_lds = rstat_lds + VGQ_STAT_SLOTS * 64;                                    // [waves][QPW queries]: (a, bb, cc, uu)
    uint64_t *pbuf_lds = reinterpret_cast<uint64_t *>(kq_lds + WAVES * QPW);             // [waves][QS sets][64]: pairs on their way out
    uint32_t *queue_lds = reinterpret_cast<uint32_t *>(pbuf_lds + WAVES * QS * 64);      // [waves][VGQ_QCAP][QENT dwords]: candidate lanes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31, h = lane >> 5;

    const int G = a.nq_pad / QPB;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int g = idx % G;
    const int part = (idx / G) * 8 + xcd;
    if (part >= a.npart) return;
    const int q0 = g * QPB + wave * QPW;                                 // set s: queries q0 + 32 s .. + 31
    const int chunks_per_row = (int)(a.stride / 16);

    // ---- A operands: lane (x, h) keeps bytes [32t + 16h, +16) of query x of each set
    vgh_i32x4 areg[QS][KS * NTB];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        const uint8_t *qrow = a.qcodes + (long long)(q0 + 32 * s + x) * a.stride;
        vgb_static_for<0, KS * NTB>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int off = 32 * t + 16 * h;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (off < a.stride) v = *reinterpret_cast<const uint4 *>(qrow + off);
            areg[s][t] = vgh_i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        });
    }
    // ---- the gates, in the query's own integer units (everything divided by its scale sq): a pair (query r, row x) passes when
    //     I sx + a_r ||ex|| + bb_r ||x|| + cc_r - [L2: m_x uu_r]  >=  0          I = qi.xi,  m_x = (1 - rel) |x|^2 / 2
    //   dot     a = ||qi||, bb = (||eq|| + rel |q|) / sq, cc = gate(thr) / sq
    //   L2      a, bb = ||eq|| / sq, cc = (gate(thr^2) - (1 - rel) |q|^2) / (2 sq), uu = 1 / sq
    //   cosine  a, bb = ||eq|| / sq - G, cc = 0, uu = G = (1 - 4e-6 - gate(thr)) |q| / sq      (a zero row passes iff G <= 0: its distance is 1.0)
    // each coefficient rounded towards "pass" by a relative 8e-6 (its own float evaluation; I sx, either sign, through a (||ex|| + ||x||)).
    // Lane l computes query q0 + l's four numbers (thresholds are fixed for the launch: the lists live in the exact-evaluation kernel);
    // the loosest of each over a query SET (the lane's half of the wavefront) feed the tile boundary's first test of that set.
    float amax[QS], bbmax[QS], ccmax[QS], uumin[QS];
    {
        const int ql = q0 + (lane & (QPW - 1));                          // (one set per wavefront: both halves compute the same 32 queries)
        const float4 s0 = a.qstat[2 * ql], s1 = a.qstat[2 * ql + 1];
        const float sq = s0.x, sqi = s0.y, eqn = s0.z, qn = s0.w, qq = s1.x;
        const float thr = a.init_keys ? vgb_kth_distance(a.init_keys[(long long)ql * 64 + (a.k - 1)]) : INFINITY;
        // (a k-th best of -Inf - a dot product against a row holding Inf - cannot be beaten: nothing passes; thr + rel |thr| would be NaN)
        const bool ok = (s1.y != 0.0f) && thr > -INFINITY;
        const float rel = a.rel, up = 1.0f + 8.0e-6f, dn = 1.0f - 8.0e-6f;
        const float u = 1.0f / sq;
        const float av = sqi * u * (up + 5.0e-6f);
        float bb = eqn * (1.0f + rel) * u * up + 5.0e-6f * av, cc = 0.0f, uu = 0.0f;
        if (COS) {
            const float gate = thr + rel * fabsf(thr) + 1e-30f;
            const float gg = 1.0f - 4.0e-6f - gate;                       // a pair passes  <=>  r + rel |r| > gg,  r = (st + E) / (|q| |x|)
            const float Gq = gg > 0.0f ? gg * qn / (1.0f + rel) * u * dn * dn : gg * qn / (1.0f - rel) * u * up * up;
            uu = Gq;
            bb = bb - Gq + 8.0e-6f * (fabsf(bb) + fabsf(Gq));
        } else if (L2M) {
            const float thr2 = a.root ? thr * thr : thr;
            const float gate2 = thr2 * (1.0f + 2.0f * rel) + 1e-30f;
            cc = 0.5f * (gate2 - (1.0f - rel) * qq) * u;
            cc += 8.0e-6f * (fabsf(cc) + (gate2 + qq) * u);
            uu = u * dn;
        } else {
            bb += rel * qn * u * up;
            cc = (thr + rel * fabsf(thr) + 1e-30f) * u;
            cc += 8.0e-6f * fabsf(cc);
        }
        // +Inf / NaN threshold (list not full, the first stage): accept everything
        if (COS) { if (!(bb < VGH_ACCEPT)) bb = VGH_ACCEPT; if (!(uu > -VGH_ACCEPT)) uu = -VGH_ACCEPT; }
        else if (!(cc < VGH_ACCEPT)) cc = VGH_ACCEPT;
        if (!ok) { cc = -VGH_ACCEPT; bb = 0.0f; uu = COS ? VGH_ACCEPT : 0.0f; }     // padding / queries the filter cannot judge: never pass
        if constexpr (PRE)                                                // (sq or 0 = not judged, sq ||qi|| up, ||eq|| + rel |q| up, |q|)
            kq_lds[wave * QPW + (lane & (QPW - 1))] = make_float4(s1.y != 0.0f ? sq : 0.0f, sqi * (1.0f + 1.0e-5f),
                                                                  (eqn * (1.0f + rel) + rel * qn) * (1.0f + 1.0e-5f), qn);
        else
        kq_lds[wave * QPW + (lane & (QPW - 1))] = make_float4(ok ? av : 0.0f, bb, cc, uu);
        float m1 = ok ? av : 0.0f, m2 = ok ? bb : -VGH_ACCEPT, m3 = ok ? cc : -VGH_ACCEPT, m4 = ok ? uu : VGH_ACCEPT;
#pragma unroll
        for (int s = 16; s >= 1; s >>= 1) {                               // over the 32 lanes of a half: one query set each
            m1 = fmaxf(m1, __shfl_xor(m1, s)); m2 = fmaxf(m2, __shfl_xor(m2, s));
            m3 = fmaxf(m3, __shfl_xor(m3, s)); m4 = fminf(m4, __shfl_xor(m4, s));
        }
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            amax[s] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m1), 32 * s));
            bbmax[s] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m2), 32 * s));
            ccmax[s] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m3), 32 * s));
            uumin[s] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m4), 32 * s));
        }
#if VGQ_STATS
        if (!PRE && part == 0 && a.tiles_per_part >= 1000 && ok) {       // how far a query's own threshold term lies from its set's loosest, relative
            const float mine = ccmax[QS > 1 ? (lane >> 5) : 0];
            const float rel1 = (mine - cc) / fabsf(cc);
            atomicAdd(&vgq_stats[30], (unsigned long long)(rel1 * 1.0e6f));
            atomicAdd(&vgq_stats[31], 1ull);
        }
#endif
    }
    const float4 *kq_w = kq_lds + wave * QPW;
    for (int s = tid; s < NB * TILE_BYTES / 4; s += THREADS) reinterpret_cast<uint32_t *>(tile0)[s] = 0u;   // pad columns
    __syncthreads();

    // ---- tile streaming by LDS-DMA (vg_batch_h.hip's FILTER kind): piece p = chunk columns 2p, 2p+1 of all 32 rows = 1 KiB of the
    // tile-major copy; wavefront w moves the contiguous pieces w * NPIECE .. + NPIECE - 1
    const int npieces = (chunks_per_row + 1) / 2;
    const long long tile_first = a.tile_begin + (long long)part * a.tiles_per_part;
    const long long tile_last = min(tile_first + a.tiles_per_part, a.tile_end);
    const int T = tile_first < tile_last ? (int)(tile_last - tile_first) : 0;
    const unsigned long long stride_b = (unsigned long long)a.stride;
    const uint32_t lds_tile0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)tile0;
    constexpr int NPIECE = (NTB + NISSUE - 1) / NISSUE;
    static_assert(NPIECE <= 4, "ring buffers of up to 512-byte (sub-)rows");
    // K-part `part` of a tile = pieces part * NTB .. + NTB - 1 of its rows (contiguous: the copy is chunk-column major within a tile)
    auto piece_mask_of = [&](int part_k, int p) -> uint64_t {
        const int gp = part_k * NTB + p;
        if (p >= NTB || gp >= npieces) return 0ull;
        return (2 * gp + 1 < chunks_per_row) ? ~0ull : 0xFFFFFFFFull;
    };
    auto full_of = [&](int part_k) -> bool { return wave * NPIECE + NPIECE <= NTB && 2 * (part_k * NTB + wave * NPIECE + NPIECE) <= chunks_per_row; };
    bool all_parts_full = true;
#pragma unroll
    for (int kp = 0; kp < KS; ++kp) all_parts_full = all_parts_full && full_of(kp);
    const uint32_t lane_goff = (uint32_t)lane * 16u;
    const uint32_t lds_rstat0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)rstat_lds;
    auto dma_stat_group = [&](long long tile2, int slot) {          // the row statistics of tiles tile2, tile2 + 1: 1 KiB
        const uint8_t *b0 = reinterpret_cast<const uint8_t *>(a.rstat + tile2 * VGQ_TILE);
        const uint32_t d0 = lds_rstat0 + (uint32_t)(slot * 1024);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_goff), "s"(b0), "s"(d0) : "memory");
    };
    auto dma_piece = [&](long long tile, int part_k, int buf, int i) __attribute__((always_inline)) {
        const int p = wave * NPIECE + i;
        const uint64_t pmask = piece_mask_of(part_k, p);
        if (pmask == 0) return;
        const uint8_t *sbase = a.rows + (unsigned long long)(tile * VGQ_TILE) * stride_b + (unsigned)(part_k * NTB + p) * 1024u;
        const uint32_t lds_dst = lds_tile0 + (uint32_t)(buf * TILE_BYTES + p * 1024);
        uint32_t keep;
        uint64_t keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_and_b64 exec, exec, %5\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keep_exec) : "v"(lane_goff), "s"(sbase), "s"(lds_dst), "s"(pmask) : "memory", "scc");
    };
    auto dma_run = [&](long long tile, int part_k, int buf) __attribute__((always_inline)) {           // NPIECE (1 .. 4) whole pieces, back to back
        const uint8_t *sbase = a.rows + (unsigned long long)(tile * VGQ_TILE) * stride_b + (unsigned)(part_k * NTB + wave * NPIECE) * 1024u;
        const uint32_t lds_dst = lds_tile0 + (uint32_t)(buf * TILE_BYTES + (wave * NPIECE) * 1024);
        uint32_t keep;
        if constexpr (NPIECE == 1)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
        else if constexpr (NPIECE == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
        else if constexpr (NPIECE == 3)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    // ring trip u = K-part u % KS of tile tile_first + u / KS
    auto dma_share = [&](int u, int buf) __attribute__((always_inline)) {
        const int part_k = KS == 1 ? 0 : u % KS;
        const long long tile = tile_first + (KS == 1 ? u : u / KS);
        if (KS == 1 ? all_parts_full : full_of(part_k)) dma_run(tile, part_k, buf);
        else vgb_static_for<0, NPIECE>([&](auto pc) { dma_piece(tile, part_k, buf, decltype(pc)::value); });
    };

    // The row statistics (16 bytes per row) ride the same queue, two tiles = 1 KiB per instruction into a ring of VGQ_STAT_SLOTS groups,
    // the wavefronts taking turns.  Group j (tiles 2j, 2j + 1 of this partition) is issued in trip 2j - NB, IN FRONT of that trip's
    // pieces: a wavefront's loads return in order and its tile-end wait leaves at most (NB - 2) * NPIECE of them outstanding, so by the
    // barrier of trip 2j - 3 at the latest the group has landed - without any wait of its own (waiting for it on the spot meant waiting
    // for the pieces issued in the same trip: a full memory round trip every other tile, for all four wavefronts at the barrier).
    // (K-parts: a group is issued LEAD = 2 tiles = 2 KS trips ahead - more than the NB - 2 trips the counted wait may leave outstanding)
    constexpr int LEAD = KS == 1 ? NB : 2;                           // tiles between a group's issue and its first use
    constexpr int SPRE = (LEAD + 1) / 2;                             // groups loaded up front: j with 2j - LEAD < 0
    const int U = T * KS;                                            // ring trips of this partition
    if (T > 0) {
        if (wave < SPRE && 2 * wave < T + 1) dma_stat_group(tile_first + 2 * wave, wave);
        vgb_static_for<0, NB - M>([&](auto jc) {                     // trips 0 .. NB - M - 1 into buffers 0 .. NB - M - 1
            constexpr int j = decltype(jc)::value;
            dma_share(min(j, U - 1), j);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool counted_wait = all_parts_full;
    const long long region0 = ((long long)(g * a.npart_total + a.part_base + part) * WAVES + wave) * QS;
    uint64_t *pairs0 = a.pairs + region0 * a.pair_cap, *pairs1 = pairs0 + (QS > 1 ? a.pair_cap : 0);
    unsigned n_pairs0 = 0, n_pairs1 = 0;                              // pairs in the regions so far (wave-uniform)
    // Pairs are collected in LDS (64 per set and wavefront) and go out 64 at a time: a store issued in the middle of the streaming loop
    // sits in the same in-order queue as the LDS-DMA pieces, and the tile-end wait "at most N outstanding" then waits for IT - a
    // ~2 us write acknowledgement - before it can count the pieces behind it as landed (one store per passing pair: 6.7 ms per batch
    // against 2.8 ms with the stores compiled out, docs/ROUND5_NOTEBOOK.md §2).
    uint64_t *pbuf0 = pbuf_lds + wave * (QS * 64), *pbuf1 = pbuf0 + (QS > 1 ? 64 : 0);
    unsigned n_buf0 = 0, n_buf1 = 0;                                  // (wave-uniform)
    auto flush = [&](uint64_t *buf, unsigned &n_buf, uint64_t *region, unsigned &n_pairs) __attribute__((always_inline)) {
        if ((unsigned)lane < n_buf && n_pairs + (unsigned)lane < (unsigned)a.pair_cap) region[n_pairs + lane] = buf[lane];
        if (n_pairs + n_buf > (unsigned)a.pair_cap && lane == 0) a.pair_counts[a.flag_index] = 1u;     // region full: the host answers the batch another way
        n_pairs += n_buf;
        n_buf = 0;
    };
    const float mfac = 0.5f * (1.0f - a.rel) * (1.0f - 8.0e-6f);
    // the lane's two sets' gate coefficients (set = h for the lane's own accumulators: register r of lane (x, h) is query (r&3) + 8 (r>>2) + 4h
    // of BOTH sets - acc0 holds set 0, acc1 set 1 - so every lane needs both sets' maxima)
    // ---- the candidate queue.  A lane whose largest accumulator passes the first test is a CANDIDATE LANE: one row, 32 queries.  Looking at
    // its 32 accumulators on the spot - which registers, then the query's own coefficients for each - was ~200 instructions and several
    // LDS round trips of the whole wavefront for, on average, 1.3 such lanes (a quarter of all tiles: + 50 % on the streaming loop,
    // docs/ROUND5_NOTEBOOK.md §2).  Instead the lane parks its accumulators, its row's statistics and its two thresholds in LDS (ten 16-byte writes)
    // and the loop goes on; every VGQ_QCAP entries the wavefront looks at them together - lane (e, o) takes registers 8 o .. 8 o + 7 of
    // entry e, all 64 lanes busy, eight steps - and the pairs that pass go to the pair buffers, entries ascending: a query's rows stay
    // in scan order.
#if VGQ_STATS
    unsigned st_slow = 0, st_cand = 0, st_pairs = 0, st_lanes = 0;
#endif
    uint32_t *queue_w = queue_lds + wave * (VGQ_QCAP * QENT);
    unsigned n_q = 0, q_head = 0;                                     // (wave-uniform) entries in the queue; the oldest one's slot (a ring)
    auto process_queue = [&]() __attribute__((always_inline)) {
        constexpr int RPL = 4 * QS;                                   // accumulator registers of an entry per lane (four lanes per entry)
        const int e = lane >> 2, o = lane & 3, set = QS > 1 ? o >> 1 : 0;
        const uint32_t *ent = queue_w + ((q_head + (unsigned)e) & (unsigned)(VGQ_QCAP - 1)) * QENT;
        const uint4 m0 = *reinterpret_cast<const uint4 *>(ent + 16 * QS), m1 = *reinterpret_cast<const uint4 *>(ent + 16 * QS + 4);
        const uint4 a0 = *reinterpret_cast<const uint4 *>(ent + RPL * o), a1 = *reinterpret_cast<const uint4 *>(ent + RPL * o + (QS > 1 ? 4 : 0));
        const uint32_t row_e = m0.x;
        const bool live = (unsigned)e < n_q && (long long)row_e < a.n_rows;      // (rows behind the corpus' end in its last tile read as zero rows)
        const int ithr_e = (int)(set ? m0.z : m0.y);
        const float sx_e = __uint_as_float(m0.w), rx_e = __uint_as_float(m1.x), nx_e = __uint_as_float(m1.y);
        const int h_e = (int)m1.z;
        const float mx2_e = L2M ? mfac * nx_e * nx_e : 0.0f;
        const int acc8[8] = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        const unsigned long long set0_lanes = QS > 1 ? 0x3333333333333333ull : ~0ull;
        // the eight queries' coefficients first, in one burst of LDS reads: read one by one inside the loop below, each iteration waited for
        // its own round trip - a queue run took ~3 700 cycles (profiles/r10_q8_tile_trace_one_workgroup.txt), and seven wavefronts wait for it
        float4 kqv[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = (QS > 1 ? (o & 1) * 8 : o * 4) + i;
            kqv[i] = kq_w[32 * set + (r & 3) + 8 * (r >> 2) + 4 * h_e];
        }
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = (QS > 1 ? (o & 1) * 8 : o * 4) + i;         // register within the set
            const int qi = (r & 3) + 8 * (r >> 2) + 4 * h_e;
            const float4 kq = kqv[i];
            const int I = acc8[i];
            bool pass = live && I >= ithr_e;
            if (nx_e < 0.0f) pass = pass && kq.z > -1.0e38f;                                       // a row that is never judged: every judged query
            else if (COS && nx_e == 0.0f) pass = pass && kq.z > -1.0e38f && kq.w <= 0.0f;          // a zero row's cosine distance is 1.0 (distance-cpu.c:74-110)
            else {
                float lhs = fmaf((float)I, sx_e, fmaf(kq.x, rx_e, fmaf(kq.y, nx_e, kq.z)));
                if constexpr (L2M) lhs = fmaf(-mx2_e, kq.w, lhs);
                pass = pass && lhs >= 0.0f;
            }
            if (VGQ_ABLATE == 5) pass = false;
            const unsigned long long pm = __ballot(pass);
#if VGQ_STATS
            st_cand += (unsigned)__popcll(__ballot(live && I >= ithr_e));
            st_pairs += (unsigned)__popcll(pm);
#endif
            if (pm == 0ull) continue;
            const unsigned long long pm0 = pm & set0_lanes, pm1 = pm & ~set0_lanes;
            const unsigned c0 = (unsigned)__popcll(pm0), c1 = (unsigned)__popcll(pm1);
            if (n_buf0 + c0 > 64u) flush(pbuf0, n_buf0, pairs0, n_pairs0);
            if (n_buf1 + c1 > 64u) flush(pbuf1, n_buf1, pairs1, n_pairs1);
            const unsigned long long mine = set ? pm1 : pm0;
            const unsigned at = (set ? n_buf1 : n_buf0) + __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0u));
            if (pass) (set ? pbuf1 : pbuf0)[at] = ((uint64_t)(uint32_t)qi << 32) | row_e;
            n_buf0 += c0; n_buf1 += c1;
        }
        n_q = 0; q_head = 0;
    };
    constexpr int BP = VGQ_BPIPE < NTB ? VGQ_BPIPE : NTB;
    vgh_i32x4 bq[BP];
    int cur_buf = 0;
    // ---- tile boundary of tile ti over the accumulators acc0 / acc1.  First test, per query set: the lane's LARGEST accumulator of the set
    // against the set's loosest gate, as an integer: I >= ithr = (-(amax ||ex|| + bbmax ||x|| + ccmax - m uumin)) / sx, rounded down.  Only a
    // set with a lane that passes looks at single accumulators (the same integer comparison, eight registers at a time), and only those
    // pairs get the query's own four coefficients.
    auto stat_of = [&](int ti) __attribute__((always_inline)) -> float4 { return rstat_lds[((ti >> 1) & (VGQ_STAT_SLOTS - 1)) * 64 + (ti & 1) * 32 + x]; };
    // (first test: the lanes with a candidate, and the two sets' integer thresholds; candidate path: those lanes park their accumulators)
    auto boundary_first = [&](const float4 rs, const vgq_i32x16 &acc0, const vgq_i32x16 &acc1, int (&ithr)[2]) __attribute__((always_inline)) -> unsigned long long {
        if (VGQ_ABLATE >= 2) asm volatile("" :: "v"(acc0[0]), "v"(acc0[15]), "v"(acc1[0]), "v"(acc1[15]));
        const float sx = rs.x, rx = rs.y, nx = rs.z;
        // 1 / sx from the statistics (a zero row, and the zero-filled statistics behind the corpus' last row: +Inf - the accumulators are all 0
        // and the sign of `rest` decides)
        const float inv_sx = sx != 0.0f ? rs.w : INFINITY;
        const float mx2 = L2M ? mfac * nx * nx : 0.0f;
        ithr[0] = 0x7FFFFFFF; ithr[1] = 0x7FFFFFFF;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            // I >= ithr  <=  I sx + rest >= 0;  ithr = floor(-(rest + sx) / sx) less a relative 1e-6: the "- 1" of the rounding rides in `rest`,
            // the float -> int conversion saturates (an open gate: -2^31, a gate nothing passes: 2^31 - 1)
            float rest = fmaf(amax[s], rx, fmaf(bbmax[s], nx, COS ? sx : ccmax[s] + sx));
            if constexpr (L2M) rest = fmaf(-mx2, uumin[s], rest);
            if (COS && nx == 0.0f) rest = -uumin[s];                   // a zero row's cosine distance is 1.0 whatever the query (distance-cpu.c:74-110)
            const float tf = -rest * inv_sx;
            // (a set of padding / unjudged queries only: rest = -Inf, tf = +Inf - and Inf - Inf below was NaN, which the conversion turns into
            //  0: every lane of every tile became a candidate of such a set, and a 300-query batch - 212 padding slots - took 7.1 ms where 512
            //  queries take 2.5; found in round 6 with ragged batch sizes, tools/r6_small_batch.py)
            const float tf2 = fabsf(tf) < 3.0e38f ? fmaf(fabsf(tf), -1.0e-6f, tf) : tf;
            int it;
            asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(it) : "v"(tf2));
            ithr[s] = it;
        }
        if (VGQ_ABLATE < 2) {
            // group maxima: registers 0-7 and 8-15 of each set
            int gm[4];
            gm[0] = max(max(max(acc0[0], acc0[1]), max(acc0[2], acc0[3])), max(max(acc0[4], acc0[5]), max(acc0[6], acc0[7])));
            gm[1] = max(max(max(acc0[8], acc0[9]), max(acc0[10], acc0[11])), max(max(acc0[12], acc0[13]), max(acc0[14], acc0[15])));
            bool cand_lane = max(gm[0], gm[1]) >= ithr[0];
            if constexpr (QS > 1) {
                gm[2] = max(max(max(acc1[0], acc1[1]), max(acc1[2], acc1[3])), max(max(acc1[4], acc1[5]), max(acc1[6], acc1[7])));
                gm[3] = max(max(max(acc1[8], acc1[9]), max(acc1[10], acc1[11])), max(max(acc1[12], acc1[13]), max(acc1[14], acc1[15])));
                cand_lane = cand_lane || max(gm[2], gm[3]) >= ithr[1];
            }
            unsigned long long m = __ballot(cand_lane);
            if (VGQ_ABLATE == 1) { asm volatile("" :: "s"(m)); m = 0ull; }
            return m;
        }
        return 0ull;
    };
    auto boundary_cands = [&](int ti, unsigned long long m, const float4 rs, const int (&ithr)[2], const vgq_i32x16 &acc0, const vgq_i32x16 &acc1) __attribute__((always_inline)) {
        const long long row_cur = (tile_first + ti) * VGQ_TILE + x;
        const float sx = rs.x, rx = rs.y, nx = rs.z;
        {
#if VGQ_STATS
            if (m) { ++st_slow; st_lanes += (unsigned)__popcll(m); }
#endif
            while (m) {                                                  // (one trip unless the queue runs full)
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const bool take = ((m >> lane) & 1ull) != 0ull && n_q + rank < (unsigned)VGQ_QCAP;
                if (take && VGQ_ABLATE != 7) {
                    uint32_t *ent = queue_w + ((q_head + n_q + rank) & (unsigned)(VGQ_QCAP - 1)) * QENT;
                    vgb_static_for<0, 4>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        *reinterpret_cast<uint4 *>(ent + 4 * j) = make_uint4((uint32_t)acc0[4 * j], (uint32_t)acc0[4 * j + 1], (uint32_t)acc0[4 * j + 2], (uint32_t)acc0[4 * j + 3]);
                        if constexpr (QS > 1)
                            *reinterpret_cast<uint4 *>(ent + 16 + 4 * j) = make_uint4((uint32_t)acc1[4 * j], (uint32_t)acc1[4 * j + 1], (uint32_t)acc1[4 * j + 2], (uint32_t)acc1[4 * j + 3]);
                    });
                    *reinterpret_cast<uint4 *>(ent + 16 * QS) = make_uint4((uint32_t)row_cur, (uint32_t)ithr[0], (uint32_t)ithr[1], __float_as_uint(sx));
                    *reinterpret_cast<uint4 *>(ent + 16 * QS + 4) = make_uint4(__float_as_uint(rx), __float_as_uint(nx), (uint32_t)h, 0u);
                }
                const unsigned long long taken = __ballot(take);
                n_q += (unsigned)__popcll(taken);
                m &= ~taken;
                if (n_q == (unsigned)VGQ_QCAP) { if (VGQ_ABLATE == 7 || VGQ_ABLATE == 8) { n_q = 0; q_head = 0; } else process_queue(); }
            }
        }
    };
    auto boundary = [&](int ti, const vgq_i32x16 &acc0, const vgq_i32x16 &acc1) __attribute__((always_inline)) {
        const float4 rs = stat_of(ti);
        int ithr[2];
        const unsigned long long m = boundary_first(rs, acc0, acc1, ithr);
        boundary_cands(ti, m, rs, ithr, acc0, acc1);
    };
    // ---- PRE: what a tile says about every query's k-th best WITHOUT an exact evaluation.  The bound that lets the filter reject a pair
    // also bounds a pair's distance from ABOVE: q.x >= sq sx I - sq ||qi|| ||ex|| - ||eq|| ||x|| =: s_lo, so the distance the exact
    // evaluation would compute for (query, row) is at most U(s_lo) (dot: -s_lo; L2: |q|^2 + |x|^2 - 2 s_lo; cosine: 1 - s_lo / (|q| |x|)),
    // every float step rounded towards "larger".  The smallest U over a tile's 32 rows belongs to ONE row; the k-th smallest of a query's
    // tile minima (vg_q8_pre_select_kernel) therefore stands for k different rows whose exact distances are no larger: a valid start
    // threshold for the first real stage - without the five warm-up stages (two tiles with every gate open, then x8, x8, x8, x8) that cost
    // 1.0 of a 1024 x 10M x 384 batch's 5.2 ms (profiles/r10_q8_stage_timeline_before.txt).
    auto pre_boundary = [&](int ti, const vgq_i32x16 &acc0, const vgq_i32x16 &acc1) __attribute__((always_inline)) -> float {
        const long long tile = tile_first + ti;
        const long long row_cur = tile * VGQ_TILE + x;
        const float4 rs = rstat_lds[((ti >> 1) & (VGQ_STAT_SLOTS - 1)) * 64 + (ti & 1) * 32 + x];
        const float sx = rs.x, rx = rs.y, nx = rs.z;
        const bool row_ok = row_cur < a.n_rows && nx >= 0.0f;            // (a row that is never judged: no witness)
        const float rel = a.rel;
        float keep = INFINITY;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float4 kq = kq_w[32 * s + qi];                      // (sq, A, B, |q|): one address per half - a broadcast read
                const int I = s ? acc1[r] : acc0[r];
                const float t = (float)I * sx;
                const float E = fmaf(kq.y, rx, kq.z * nx);
                float s_lo = fmaf(kq.x, t, -E);
                s_lo -= 1.0e-5f * (fabsf(kq.x * t) + E) + 1.0e-30f;
                float U;
                if constexpr (COS) {
                    const float den = kq.w * nx;
                    const float ratio = s_lo >= 0.0f ? s_lo / (den * (1.0f + 4.0f * rel)) : s_lo * (1.0f + 4.0f * rel) / den;
                    U = nx == 0.0f ? 1.0f : fminf(1.0f - ratio + 8.0e-6f + 2.0f * rel, 2.0f);
                } else if constexpr (L2M) {
                    const float qq = kq.w * kq.w * (1.0f + 1.0e-6f);
                    float U2 = fmaf(-2.0f, s_lo, (qq + nx * nx) * (1.0f + 4.0f * rel));
                    U2 = fmaxf(U2 * (1.0f + 8.0f * rel), 0.0f) + 1.0e-30f;
                    U = (a.root ? sqrtf(U2) : U2) * (1.0f + 1.0e-6f);
                } else {
                    U = -s_lo;
                    U += 1.0e-6f * fabsf(U);
                }
                if (fabsf(U) < 1.0e-5f) U = 1.0e-5f;                      // (nearly_zero_float32 moves a tiny distance to 0: sqlite-vector.c:994-996)
                if (!(row_ok && kq.x > 0.0f) || !(U == U)) U = INFINITY;
                // the smallest over the half's 32 lanes = the tile's 32 rows
                uint32_t ub = __float_as_uint(U);
                auto fmin_u = [](uint32_t p, uint32_t q) { return __float_as_uint(fminf(__uint_as_float(p), __uint_as_float(q))); };
                ub = fmin_u(ub, vg_dpp_u32<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(ub));
                ub = fmin_u(ub, vg_dpp_u32<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(ub));
                ub = fmin_u(ub, vg_dpp_u32<VG_DPP_ROW_HALF_MIRROR>(ub));
                ub = fmin_u(ub, vg_dpp_u32<VG_DPP_ROW_MIRROR>(ub));
                ub = fmin_u(ub, (uint32_t)__shfl_xor((int)ub, 16));
                if ((x & 15) == r && (x >> 4) == s) keep = __uint_as_float(ub);
            }
        }
        return keep;                                                      // lane (x, h) keeps query 32 (x >> 4) + qi(x & 15, h) of the wavefront
    };
    // (the minima leave BEHIND a group's barrier: a store in front of it sits in the LDS-DMA's in-order queue and the group-end wait then
    //  waits for its acknowledgement - 116 us for 2 048 tiles against ~40)
    float pre_keep[M > 0 ? M : 1];
    auto pre_store = [&](int ti_first, int count) __attribute__((always_inline)) {
        if ((x >> 4) < QS) {
            const int r = x & 15;
            const int qw = 32 * (x >> 4) + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < count) a.premin[(tile_first + ti_first + j - a.tile_begin) * (long long)a.nq_pad + q0 + qw] = pre_keep[j];
        }
    };
    // SOFTWARE-PIPELINED BOUNDARY (VGQ_SWP, the short-row forms - 256 queries per workgroup too: 1.60 -> 1.575 / 2.07 -> 1.98 ms per 256-query batch): a wavefront owns TWO pairs of accumulator sets; tile t's MFMAs run into one pair
    // while the first test of tile t - 1 (thresholds from the row statistics, the maxima, the ballot: ~55 VALU instructions, no memory access)
    // reads the other, in the SAME instruction stream - the matrix pipe takes 32 cycles per MFMA and the issue port 4, so up to ~5 other
    // instructions fit beside every MFMA (MI355X_MICROARCH.md) instead of a block of VALU work during which this wavefront offers the pipe nothing.
    constexpr bool SWP = VGQ_SWP != 0 && !PRE && KS == 1 && QS == 2 && NTB <= 12;    // (both short-row forms: eight wavefronts x 512 queries, four x 256)     // (16 k-steps: the second pair spills)
    vgq_i32x16 acc0, acc1, accb0, accb1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; accb0[r] = 0; accb1[r] = 0; }
    float4 rs_prev = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                 // (SWP) the previous tile's row statistics
#if VGQ_TIMING
    unsigned long long tk_k = 0, tk_b = 0, tk_w = 0, tk_v = 0;
    const unsigned long long tk_loop0 = __builtin_readcyclecounter();
#endif
    // one tile: its MFMAs into (acc0, acc1); SWP: the previous tile's accumulators are (old0, old1)
    // DRAIN (SWP, VGQ_SWP_DRAIN): a queue run - 16 parked candidate lanes looked at in one go, ~3 700 cycles - holds up ONE wavefront, and the other
    // seven then wait for it at the group's barrier: with a run every ~6 tiles somewhere in the workgroup that was most of the ~0.6 ms the
    // candidates cost inside this kernel.  Instead a tile whose wavefront has parked lanes retires the two OLDEST of them beside its MFMAs,
    // branch-free: lane (e, set, r) takes accumulator r of set `set` of entry e - five LDS reads issued behind k-step 0 (counted waits: they
    // ride the B operands' in-order queue), the pair's own test and the compaction into the pair buffers in the last two k-steps.  Runs
    // remain for a queue that fills faster than that (the first stages) and for the partition's end.
    constexpr bool CAN_DRAIN = SWP && VGQ_SWP_DRAIN != 0 && NTB >= 8 && VGQ_BPIPE + 3 <= NTB;
    auto tile_step = [&](int ti, vgq_i32x16 &acc0, vgq_i32x16 &acc1, const vgq_i32x16 &old0, const vgq_i32x16 &old1, auto drain_c) __attribute__((always_inline)) {
        constexpr bool DRAIN = decltype(drain_c)::value;
        const int sj = (ti + LEAD) >> 1;                                      // the statistics group this tile's first trip may issue
        const bool stat_turn = ((ti + LEAD) & 1) == 0 && 2 * sj < T + 1 && wave == (sj & (WAVES - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
        int ithr_prev[2] = {0x7FFFFFFF, 0x7FFFFFFF};
        unsigned long long m_prev = 0ull;
        // one ring trip per K-part of the tile (KS = 1: the whole row): the accumulators run on across the parts
        vgb_static_for<0, KS>([&](auto kc) {
        constexpr int kp = decltype(kc)::value;
        VGQ_TICK(tk0);
#if VGQ_TRACE
        const unsigned long long tr_k0 = __builtin_readcyclecounter();
#endif
        const int fill_buf = cur_buf + NB - M >= NB ? cur_buf - M : cur_buf + NB - M;   // (a buffer of the previous group)
        const int u_next = min(ti * KS + kp + NB - M, U - 1);                 // the trip whose DMA this trip issues
        const uint32_t baddr = lds_tile0 + (uint32_t)(cur_buf * TILE_BYTES + h * 512 + x * 16);
        if constexpr (SWP && VGQ_ABLATE < 3) {
            // (the DMA issue has branches: in front of the k loop, so that the MFMAs and the previous tile's first test share ONE basic block)
            if (stat_turn) dma_stat_group(tile_first + 2 * sj, sj & (VGQ_STAT_SLOTS - 1));
            dma_share(u_next, fill_buf);
        }
        // (SWP) the previous tile's first test in pieces of <= 5 VALU instructions, one piece per k-step (the same arithmetic as boundary_first)
        float sw_tf[2] = {0.0f, 0.0f};
        int sw_gm[4] = {0, 0, 0, 0};
        auto first_piece = [&](auto pc) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            const float sx = rs_prev.x, rx = rs_prev.y, nx = rs_prev.z;
            if constexpr (p == 0 || p == 1) {
                const float inv_sx = sx != 0.0f ? rs_prev.w : INFINITY;
                float rest = fmaf(amax[p], rx, fmaf(bbmax[p], nx, COS ? sx : ccmax[p] + sx));
                if constexpr (L2M) rest = fmaf(-(mfac * nx * nx), uumin[p], rest);
                if (COS && nx == 0.0f) rest = -uumin[p];
                sw_tf[p] = -rest * inv_sx;
            } else if constexpr (p == 2 || p == 3) {
                const float tf = sw_tf[p - 2];
                const float tf2 = fabsf(tf) < 3.0e38f ? fmaf(fabsf(tf), -1.0e-6f, tf) : tf;
                int it;
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(it) : "v"(tf2));
                ithr_prev[p - 2] = it;
            } else if constexpr (p >= 4 && p <= 7) {
                const vgq_i32x16 &o = p < 6 ? old0 : old1;
                constexpr int b = (p & 1) * 8;
                sw_gm[p - 4] = max(max(max(o[b], o[b + 1]), max(o[b + 2], o[b + 3])), max(max(o[b + 4], o[b + 5]), max(o[b + 6], o[b + 7])));
            } else if constexpr (p == 8) {
                // (no "||": the compiler turns it into a branch and sinks the second set's maxima under it)
                const int c0 = max(sw_gm[0], sw_gm[1]) >= ithr_prev[0] ? 1 : 0, c1 = max(sw_gm[2], sw_gm[3]) >= ithr_prev[1] ? 1 : 0;
                m_prev = __ballot((c0 | c1) != 0);
                if (VGQ_ABLATE == 1) { asm volatile("" :: "s"(m_prev)); m_prev = 0ull; }
                if (ti == 0) m_prev = 0ull;
            }
        };
        // (DRAIN) lane = (entry e = lane >> 5, set = (lane >> 4) & 1, register r = lane & 15)
        const unsigned n_take = DRAIN ? (n_q < 2u ? n_q : 2u) : 0u;
        const int d_e = lane >> 5, d_set = (lane >> 4) & 1, d_r = lane & 15;
        const uint32_t d_ent = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)queue_w +
                               ((q_head + (unsigned)d_e) & (unsigned)(VGQ_QCAP - 1)) * (unsigned)(QENT * 4);
        const uint32_t d_kqa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)const_cast<float4 *>(kq_w) +
                               (uint32_t)((32 * d_set + (d_r & 3) + 8 * (d_r >> 2)) * 16);
        int d_I = 0;
        vgh_i32x4 d_m0 = {0, 0, 0, 0}, d_m1 = {0, 0, 0, 0}, d_kq0 = {0, 0, 0, 0}, d_kq1 = {0, 0, 0, 0};
        vgb_static_for<0, BP>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            vgq_lds_read128<1024 * t>(bq[t], baddr);
        });
        vgb_static_for<0, NTB>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            // (DRAIN: the five entry reads sit between operand BP and operand BP + 1 in the queue - younger than operands 1 .. BP)
            constexpr int in_flight_after = ((NTB - 1 - t) < (BP - 1) ? (NTB - 1 - t) : (BP - 1)) + (DRAIN && t >= 1 && t <= BP ? 5 : 0);
            if constexpr (DRAIN && t == BP + 1)
                asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(bq[t % BP]), "+v"(d_I), "+v"(d_m0), "+v"(d_m1), "+v"(d_kq0), "+v"(d_kq1) : "n"(in_flight_after));
            else
            vgq_wait_lds<in_flight_after>(bq[t % BP]);
            const vgh_i32x4 b = bq[t % BP];
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[0][kp * NTB + t], b, acc0, 0, 0, 0);
            if constexpr (QS > 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[QS - 1][kp * NTB + t], b, acc1, 0, 0, 0);
            if constexpr (t + BP < NTB) vgq_lds_read128<1024 * (t + BP)>(bq[t % BP], baddr);
            if constexpr (DRAIN && t == 0) {
                asm volatile("ds_read_b32 %0, %1" : "=v"(d_I) : "v"(d_ent + (uint32_t)((16 * d_set + d_r) * 4)) : "memory");
                vgq_lds_read128<16 * QS * 4>(d_m0, d_ent);
                vgq_lds_read128<16 * QS * 4 + 16>(d_m1, d_ent);
                vgq_lds_read128<0>(d_kq0, d_kqa);
                vgq_lds_read128<64>(d_kq1, d_kqa);
            }
            if constexpr (DRAIN && t == NTB - 2) {
                // the pair's own test (process_queue's arithmetic) and its slot in the set's pair buffer
                const uint32_t row_e = (uint32_t)d_m0[0];
                const int ithr_e = d_set ? d_m0[2] : d_m0[1];
                const float sx_e = __int_as_float(d_m0[3]), rx_e = __int_as_float(d_m1[0]), nx_e = __int_as_float(d_m1[1]);
                const int h_e = d_m1[2];
                const vgh_i32x4 kqi = h_e ? d_kq1 : d_kq0;
                const float kqx = __int_as_float(kqi[0]), kqy = __int_as_float(kqi[1]), kqz = __int_as_float(kqi[2]), kqw = __int_as_float(kqi[3]);
                float lhs = fmaf((float)d_I, sx_e, fmaf(kqx, rx_e, fmaf(kqy, nx_e, kqz)));
                if constexpr (L2M) lhs = fmaf(-(mfac * nx_e * nx_e), kqw, lhs);
                const bool judged = kqz > -1.0e38f;
                const bool by_row = nx_e < 0.0f ? judged : ((COS && nx_e == 0.0f) ? (judged && kqw <= 0.0f) : lhs >= 0.0f);
                const bool pass = (unsigned)d_e < n_take && (long long)row_e < a.n_rows && d_I >= ithr_e && by_row;
                const unsigned long long pm = __ballot(pass);
                const unsigned long long set0_lanes = 0x0000FFFF0000FFFFull;
                const unsigned long long mine = d_set ? (pm & ~set0_lanes) : (pm & set0_lanes);
                const unsigned at = (d_set ? n_buf1 : n_buf0) + __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0u));
                const int qi = (d_r & 3) + 8 * (d_r >> 2) + 4 * h_e;
                if (pass) (d_set ? pbuf1 : pbuf0)[at] = ((uint64_t)(uint32_t)qi << 32) | row_e;
                n_buf0 += (unsigned)__popcll(pm & set0_lanes); n_buf1 += (unsigned)__popcll(pm & ~set0_lanes);
#if VGQ_STATS
                st_cand += (unsigned)__popcll(__ballot((unsigned)d_e < n_take && d_I >= ithr_e)); st_pairs += (unsigned)__popcll(pm);
#endif
            }
            if constexpr (t == 0 && VGQ_ABLATE < 3 && !SWP) {
                if (kp == 0 && stat_turn) dma_stat_group(tile_first + 2 * sj, sj & (VGQ_STAT_SLOTS - 1));
                dma_share(u_next, fill_buf);
            }
            // nothing else moves into the k loop: left alone, the compiler sinks the tile boundary's float work (thresholds from the row
            // statistics) between the MFMAs - and every extra issue slot between two MFMAs on one accumulator stalls the chain
            // (MI355X_MICROARCH.md: + 43 cycles for the first one): last stage of 1024 x 10M x 384 2.65 ms against 1.88
            if constexpr (SWP) {
                // pieces 0 .. 8 over the k-steps (NTB < 9: the rest behind the last one)
                if constexpr (t < NTB - 1) first_piece(tc);
                else { constexpr int PEND = NTB > 9 ? NTB : 9; vgb_static_for<NTB - 1, PEND>([&](auto pc) { first_piece(pc); }); }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        VGQ_TICK(tk1);
#if VGQ_TRACE
        const unsigned long long tr_k1 = __builtin_readcyclecounter();
#endif
        if constexpr (kp == KS - 1) {
            if constexpr (PRE) {
                const float kv = pre_boundary(ti, acc0, acc1);
                const int j = KS > 1 ? 0 : ti % M;                         // (K-parts: a group is one tile)
#pragma unroll
                for (int jj = 0; jj < M; ++jj) if (jj == j) pre_keep[jj] = kv;
            } else if constexpr (SWP) {
                if constexpr (DRAIN) { q_head = (q_head + n_take) & (unsigned)(VGQ_QCAP - 1); n_q -= n_take; }
                if (ti > 0) boundary_cands(ti - 1, m_prev, rs_prev, ithr_prev, old0, old1);
                if constexpr (CAN_DRAIN) {
                    // a drained tile adds up to 32 pairs per set: every tile starts with room for them (a queue run above may have left 64)
                    if (n_buf0 > 32u) flush(pbuf0, n_buf0, pairs0, n_pairs0);
                    if (n_buf1 > 32u) flush(pbuf1, n_buf1, pairs1, n_pairs1);
                }
                rs_prev = stat_of(ti);                                       // (read before the group's barrier, as the boundary itself would)
            } else boundary(ti, acc0, acc1);
        }
        VGQ_TICK(tk2);
#if VGQ_TRACE
        const unsigned long long tr_b1 = __builtin_readcyclecounter();
#endif
        // group end: the next group's pieces have landed (an issuing wavefront leaves the pieces of the (LOOK - 1) M youngest trips in
        // flight: loads return in order), barrier: every wavefront has read this group's buffers, the next group's are readable
        if (M == 1 || (ti * KS + kp) % M == M - 1) {
        if (counted_wait) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((LOOK - 1) * M * NPIECE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::: "memory");
#if VGQ_TIMING
        { const unsigned long long tkv = __builtin_readcyclecounter(); tk_v += tkv - tk2; }
#endif
        if (VGQ_ABLATE < 4) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (PRE) { if (KS > 1) pre_store(ti, 1); else pre_store(ti - (M - 1), M); }
        }
        cur_buf = cur_buf + 1 == NB ? 0 : cur_buf + 1;
#if VGQ_TRACE
        if (!PRE && KS == 1 && WAVES == 8 && T > 1000 && blockIdx.x == 37 && ti >= 200 && ti < 200 + VGQ_TRACE_TILES && lane == 0) {
            unsigned long long *tr = vgq_trace + ((size_t)wave * VGQ_TRACE_TILES + (ti - 200)) * 4;
            tr[0] = tr_k0; tr[1] = tr_k1; tr[2] = tr_b1; tr[3] = __builtin_readcyclecounter();
        }
#endif
#if VGQ_TIMING
        { const unsigned long long tk3 = __builtin_readcyclecounter(); tk_k += tk1 - tk0; tk_b += tk2 - tk1; tk_w += tk3 - tk2; }
#endif
        });
    };
    auto step = [&](int ti, vgq_i32x16 &n0, vgq_i32x16 &n1, const vgq_i32x16 &o0, const vgq_i32x16 &o1) __attribute__((always_inline)) {
        if constexpr (CAN_DRAIN) {
            if (n_q >= (unsigned)VGQ_DRAIN_MIN) tile_step(ti, n0, n1, o0, o1, std::true_type{});
            else tile_step(ti, n0, n1, o0, o1, std::false_type{});
        } else tile_step(ti, n0, n1, o0, o1, std::false_type{});
    };
    if constexpr (SWP) {
        int ti = 0;
        for (; ti + 1 < T; ti += 2) { step(ti, acc0, acc1, accb0, accb1); step(ti + 1, accb0, accb1, acc0, acc1); }
        if (ti < T) { step(ti, acc0, acc1, accb0, accb1); boundary(T - 1, acc0, acc1); }
        else if (T > 0) boundary(T - 1, accb0, accb1);
    } else
        for (int ti = 0; ti < T; ++ti) step(ti, acc0, acc1, acc0, acc1);
#if VGQ_TIMING
    if (lane == 0 && T > 64) {
        const int o = (WAVES == 8 && wave >= 4) ? 8 : 0;
        atomicAdd(&vgq_ticks[o + 0], __builtin_readcyclecounter() - tk_loop0); atomicAdd(&vgq_ticks[o + 1], tk_k); atomicAdd(&vgq_ticks[o + 2], tk_b);
        atomicAdd(&vgq_ticks[o + 3], tk_w); atomicAdd(&vgq_ticks[o + 4], (unsigned long long)T); atomicAdd(&vgq_ticks[o + 5], tk_v);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (no LDS-DMA of the ring may land after this workgroup's LDS is gone)
#if VGQ_STATS
    if (lane == 0 && !PRE) {
        const int b = 5 * (T < 16 ? 0 : (T < 64 ? 1 : (T < 256 ? 2 : (T < 448 ? 3 : (T < 1000 ? 4 : 5)))));
        atomicAdd(&vgq_stats[b + 0], (unsigned long long)T); atomicAdd(&vgq_stats[b + 1], (unsigned long long)st_slow);
        atomicAdd(&vgq_stats[b + 2], (unsigned long long)st_lanes); atomicAdd(&vgq_stats[b + 3], (unsigned long long)st_cand);
        atomicAdd(&vgq_stats[b + 4], (unsigned long long)st_pairs);
    }
#endif
    if constexpr (PRE) {
        if (KS == 1 && M > 1 && T % M != 0) pre_store(T - T % M, T % M);
        return;
    }
    if (n_q && VGQ_ABLATE != 7 && VGQ_ABLATE != 8) process_queue();
    flush(pbuf0, n_buf0, pairs0, n_pairs0);
    if (QS > 1) flush(pbuf1, n_buf1, pairs1, n_pairs1);
    if (lane == 0) {
        a.pair_counts[region0] = n_pairs0 < (unsigned)a.pair_cap ? n_pairs0 : (unsigned)a.pair_cap;
        if (QS > 1) a.pair_counts[region0 + 1] = n_pairs1 < (unsigned)a.pair_cap ? n_pairs1 : (unsigned)a.pair_cap;
    }
}

// ---- host side
template <int NTB, int MODE, int WAVES, int QS, int KS>
static int launch_q8_form(const BatchArgsQ8 &a, int blocks, size_t smem, hipStream_t stream, bool pre) {
    auto kern = pre ? vg_batch_q8_kernel<NTB, MODE, WAVES, QS, KS, true> : vg_batch_q8_kernel<NTB, MODE, WAVES, QS, KS, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WAVES), smem, stream, a);
    return (int)hipGetLastError();
}
template <int NTB, int WAVES, int QS, int KS>
static int launch_q8_form_mode(const BatchArgsQ8 &a, int blocks, size_t smem, hipStream_t stream, bool pre) {
    if (a.mode == VGH_COS) return launch_q8_form<NTB, VGH_COS, WAVES, QS, KS>(a, blocks, smem, stream, pre);
    if (a.mode == VGH_L2) return launch_q8_form<NTB, VGH_L2, WAVES, QS, KS>(a, blocks, smem, stream, pre);
    return launch_q8_form<NTB, VGH_DOT, WAVES, QS, KS>(a, blocks, smem, stream, pre);
}
template <int NTB>
static int launch_q8_mode(const BatchArgsQ8 &a, int blocks, size_t smem, hipStream_t stream, bool pre) {
    return launch_q8_form_mode<NTB, VGQ_WAVES, VGQ_QS, 1>(a, blocks, smem, stream, pre);
}
// the NARROW form (round 6): four wavefronts x ONE query set = 128 queries per workgroup, two workgroups per CU - for batches of up to 128
// queries.  The 256-query form multiplies 256 query slots whatever the batch holds, and with few real queries (no candidates to speak of) it
// runs at 96 % of the matrix pipe: a 16-query batch paid for 240 padding slots (1.4-1.5 ms).  Half the slots, half the MFMAs.
template <int NTB>
static int launch_q8n_mode(const BatchArgsQ8 &a, int blocks, size_t smem, hipStream_t stream, bool pre) {
    return launch_q8_form_mode<NTB, 4, 1, 1>(a, blocks, smem, stream, pre);
}
static size_t vgqn_lds_bytes(int NTB) {
    return (size_t)VGQ_RING_OF(NTB) * NTB * 1024 + (size_t)VGQ_STAT_SLOTS * 1024 + (size_t)4 * 32 * 16 + (size_t)4 * 64 * 8 + (size_t)4 * VGQ_QCAP * (16 + 8) * 4;
}
// the WIDE form of the short-row kernel: eight wavefronts x two query sets = 512 queries per workgroup, one workgroup per CU - a tile is
// brought in once for 512 queries instead of once per 256 (half the LDS-DMA issues and half the L2 traffic per pair), six ring buffers
template <int NTB>
static int launch_q8w_mode(const BatchArgsQ8 &a, int blocks, size_t smem, hipStream_t stream, bool pre) {
    return launch_q8_form_mode<NTB, 8, 2, 1>(a, blocks, smem, stream, pre);
}
static size_t vgqw_lds_bytes(int NTB) {
    return (size_t)VGQW_RING_OF(NTB) * NTB * 1024 + (size_t)VGQ_STAT_SLOTS * 1024 + (size_t)8 * 64 * 16 + (size_t)8 * 128 * 8 + (size_t)8 * VGQ_QCAP * 160;
}
// long rows: KS K-parts of NTB k-steps, one query set per wavefront, eight wavefronts
template <int NTB, int KS>
static int launch_q8l_mode(const BatchArgsQ8 &a, int blocks, size_t smem, hipStream_t stream, bool pre) {
    return launch_q8_form_mode<NTB, VGQL_WAVES, VGQL_QS, KS>(a, blocks, smem, stream, pre);
}
// rows of 513 .. 1536 int8 elements: (k-steps per K-part) * 8 + (K-parts), 0 if not served
static int vgql_cfg(long long stride_bytes) {
    const int ntb = (int)((stride_bytes + 31) / 32);
    if (ntb <= 16) return 0;
    if (ntb <= 24) return 12 * 8 + 2;
    if (ntb <= 32) return 16 * 8 + 2;
    if (ntb <= 48) return 16 * 8 + 3;
    return 0;
}
static size_t vgql_lds_bytes(int NTB) {
    return (size_t)VGQL_RING * NTB * 1024 + (size_t)VGQ_STAT_SLOTS * 1024 + (size_t)VGQL_WAVES * 32 * VGQL_QS * 16 + (size_t)VGQL_WAVES * VGQL_QS * 64 * 8 +
           (size_t)VGQL_WAVES * VGQ_QCAP * (16 * VGQL_QS + 8) * 4;
}
static int vgq_ntb(long long stride_bytes) {
    const int ntb = (int)((stride_bytes + 31) / 32);
    if (ntb <= 4) return 4;
    if (ntb <= 8) return 8;
    if (ntb <= 12) return 12;
    if (ntb <= 16) return 16;
    return 0;
}
static size_t vgq_lds_bytes(int NTB) {
    return (size_t)VGQ_RING_OF(NTB) * NTB * 1024 + (size_t)VGQ_STAT_SLOTS * 1024 + (size_t)VGQ_WAVES * 64 * 16 + (size_t)VGQ_WAVES * 128 * 8 + (size_t)VGQ_WAVES * VGQ_QCAP * 160;
}

extern "C" int vgh_launch_exact_f32(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream);   // vg_batch_h.hip (-DVGH_TU=8)
extern "C" int vgh_launch_exact_f16(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream);   // (-DVGH_TU=6)
extern "C" int vgh_launch_exact_bf16(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream);  // (-DVGH_TU=7)
// the same kernel with more chunks per lane (rows beyond 4 KiB f32 / 2 KiB f16 - bf16): vg_batch_hl.hip's instantiations
extern "C" int vghl_exact_regions_f16(const BatchArgsH *a, int waves, int regions, size_t smem, hipStream_t stream);
extern "C" int vghl_exact_regions_bf16(const BatchArgsH *a, int waves, int regions, size_t smem, hipStream_t stream);
extern "C" int vghl_exact_regions_f32(const BatchArgsH *a, int waves, int regions, size_t smem, hipStream_t stream);
extern "C" int vg_batch_merge_launch(const uint64_t *dev_cand, int nq_pad, int lists_per_query, int npart, int k,
                                     uint64_t *dev_out_keys, hipStream_t stream);        // vg_batch.hip

// does the int8 batch filter serve rows of q8stride_bytes int8 elements (xstride_bytes: the f32 rows) with lists of k?
extern "C" int vg_batch_q8_serves(long long q8stride_bytes, long long xstride_bytes, int k) {
    return (vgq_ntb(q8stride_bytes) != 0 || vgql_cfg(q8stride_bytes) != 0) && k >= 1 && k <= VGQ_MAX_K && xstride_bytes <= 6144;
}
// workgroups of the filter kernel a CU holds at once (what the caller sizes the partition count by)
extern "C" int vg_batch_q8_workgroups_per_cu(long long q8stride_bytes) { return vgq_ntb(q8stride_bytes) != 0 ? 2 : 1; }
extern "C" int vg_batch_q8_queries_per_block(void) { return VGQ_QPB; }
// what a batch of nq queries is padded to: short rows in the narrow form (128 query slots) up to 128 queries, whole 256-query workgroups otherwise
extern "C" int vg_batch_q8_max_partitions(void) { return VGQ_NARROW_PARTS; }        // (the 128-slot form; every other form: VG_SEL_MAX_HEADS)
extern "C" int vg_batch_q8_padded_queries(int nq, long long q8stride_bytes) {
    if (VGQ_NARROW != 0 && nq <= 128 && vgq_ntb(q8stride_bytes) != 0) return 128;
    return ((nq + VGQ_QPB - 1) / VGQ_QPB) * VGQ_QPB;
}
extern "C" int vg_batch_q8_max_queries(void) { return 4096; }         // (the rank kernel's LDS)
extern "C" int vg_batch_q8_regions(int nq_pad, int npart) { return (nq_pad / 32) * npart; }
// scratch behind the nq_pad query rows the caller uploads: the sorted query rows, their int8 images, statistics, sort keys + own scales +
// the permutation + the common scale
extern "C" size_t vg_batch_q8_work_bytes(int nq_pad, long long q8stride_bytes, long long xstride_bytes) {
    return (size_t)nq_pad * xstride_bytes + (size_t)nq_pad * q8stride_bytes + (size_t)nq_pad * 2 * sizeof(float4) + (size_t)nq_pad * 12 + 16 +
           (size_t)VGQ_PRE_TILES * nq_pad * sizeof(float);              // (+ the pre-pass' tile minima)
}
// where the statistics (nq_pad x 32 bytes) and the permutation (nq_pad ints: slot -> query) sit in that scratch
extern "C" size_t vg_batch_q8_work_stat_offset(int nq_pad, long long q8stride_bytes, long long xstride_bytes) {
    return (size_t)nq_pad * xstride_bytes + (size_t)nq_pad * q8stride_bytes;
}
extern "C" size_t vg_batch_q8_work_perm_offset(int nq_pad, long long q8stride_bytes, long long xstride_bytes) {
    return vg_batch_q8_work_stat_offset(nq_pad, q8stride_bytes, xstride_bytes) + (size_t)nq_pad * 2 * sizeof(float4) + (size_t)nq_pad * 8;
}

// ---- what the host needs of a finished batch, in ONE buffer and one copy: [0] the overflow flag, [1] exact evaluations so far (64 bit: words
// 2, 3), then per slot p: perm[p], judged[p], then the k keys of every slot.  (Five separate copies - three of them into pageable memory -
// cost ~0.2 ms of host latency per batch: profiles/r10_q8_stage_timeline_*.txt.)
__global__ __launch_bounds__(256) void vg_q8_pack_kernel(const uint64_t *keys, const int *perm, const float4 *qstat, const uint32_t *overflow_flag,
                                                         const unsigned long long *evals, int nq_pad, int k, uint32_t *out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p == 0) { out[0] = *overflow_flag; out[1] = 0u; const unsigned long long e = evals ? *evals : 0ull; out[2] = (uint32_t)e; out[3] = (uint32_t)(e >> 32); }
    if (p >= nq_pad) return;
    out[4 + p] = (uint32_t)perm[p];
    out[4 + nq_pad + p] = qstat[2 * p + 1].y != 0.0f ? 1u : 0u;
    uint64_t *ko = reinterpret_cast<uint64_t *>(out + 4 + 2 * nq_pad) + (long long)p * k;
    for (int j = 0; j < k; ++j) ko[j] = keys[(long long)p * 64 + j];
}
extern "C" size_t vg_batch_q8_pack_bytes(int nq_pad, int k) { return 16 + (size_t)nq_pad * 8 + (size_t)nq_pad * k * 8; }
extern "C" int vg_batch_q8_pack_launch(const uint64_t *dev_keys, const void *dev_qwork, int nq_pad, long long q8stride, long long xstride, int k,
                                       const uint32_t *dev_overflow_flag, const unsigned long long *dev_evals, void *dev_out, hipStream_t stream) {
    const uint8_t *w = reinterpret_cast<const uint8_t *>(dev_qwork);
    const float4 *qstat = reinterpret_cast<const float4 *>(w + vg_batch_q8_work_stat_offset(nq_pad, q8stride, xstride));
    const int *perm = reinterpret_cast<const int *>(w + vg_batch_q8_work_perm_offset(nq_pad, q8stride, xstride));
    hipLaunchKernelGGL(vg_q8_pack_kernel, dim3((unsigned)((nq_pad + 255) / 256)), dim3(256), 0, stream, dev_keys, perm, qstat, dev_overflow_flag, dev_evals,
                       nq_pad, k, reinterpret_cast<uint32_t *>(dev_out));
    return (int)hipGetLastError();
}

extern "C" int vg_q8_rstat_launch(const void *dev_q8stat, const float *dev_xnorm, long long row0, long long n, void *dev_out, int nn_squared, hipStream_t stream) {
    if (n <= 0) return 0;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(vg_q8_rstat_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float2 *>(dev_q8stat), dev_xnorm, row0, n,
                       reinterpret_cast<float4 *>(dev_out), nn_squared);
    return (int)hipGetLastError();
}

// The whole batch: sort + query images -> staged passes over growing row ranges (int8 filter -> pairs -> exact evaluation -> merge), the
// first of them two tiles wide with every gate open.
// dev_rows_tm / dev_rstat: the tile-major int8 copy + its statistics; dev_xrows: the f32 corpus (xstride bytes per row); dev_xqueries: the
// f32 queries, zero padded to nq_pad rows of xstride bytes; dev_qwork: vg_batch_q8_work_bytes() of scratch; dev_cand: npart lists of 64
// keys per query; dev_pairs: vg_batch_q8_regions() x pair_cap pairs, dev_pair_counts: regions + 1 words.  On return dev_out_keys holds the
// merged lists (nq_pad x 64) IN SORTED SLOT ORDER (slot p = query perm[p], the permutation is in the scratch) and
// dev_pair_counts[regions] the overflow flag.  Returns 0, -1 if the shape is not served, a hipError_t otherwise.
extern "C" int vg_batch_q8_launch(const uint8_t *dev_rows_tm, const void *dev_rstat, long long n_rows, long long q8stride, int dim,
                                  const uint8_t *dev_xrows, long long xstride, const float *dev_xnorm,
                                  const uint8_t *dev_xqueries, void *dev_qwork, int nq_pad, int nq_real, int k, int mode, int root,
                                  uint64_t *dev_cand, int npart, uint64_t *dev_out_keys, unsigned long long *dev_evals,
                                  uint64_t *dev_pairs, uint32_t *dev_pair_counts, int pair_cap, int type_code, hipStream_t stream) {
    const int lcfg = vgql_cfg(q8stride);
    const int ntb = lcfg ? lcfg / 8 : vgq_ntb(q8stride), ks = lcfg ? lcfg % 8 : 1;
    if (type_code < 0 || type_code > 2) return -1;
    if (!ntb || !vg_batch_q8_serves(q8stride, xstride, k) || (nq_pad % VGQ_QPB != 0 && !(nq_pad == 128 && !lcfg)) || nq_pad > vg_batch_q8_max_queries() || npart < 8 || npart % 8 != 0 ||
        npart > ((nq_pad <= 256 && !lcfg) ? VGQ_NARROW_PARTS : VG_SEL_MAX_HEADS)) return -1;
    if (mode < VGH_DOT || mode > VGH_L2 || !dev_xnorm || !dev_pairs || !dev_pair_counts || pair_cap < 32 * 32) return -1;
    const long long ntiles = (n_rows + VGQ_TILE - 1) / VGQ_TILE;
    if (ntiles < 2048) return -1;                                     // small corpora: the lists warm up inside a fused kernel instead
    uint8_t *xq_sorted = reinterpret_cast<uint8_t *>(dev_qwork);
    uint8_t *qcodes = xq_sorted + (size_t)nq_pad * xstride;
    float4 *qstat = reinterpret_cast<float4 *>(qcodes + (size_t)nq_pad * q8stride);
    float *qkeys = reinterpret_cast<float *>(qstat + (size_t)nq_pad * 2);
    float *qscales = qkeys + nq_pad;
    int *perm = reinterpret_cast<int *>(qscales + nq_pad);
    float *common = reinterpret_cast<float *>(perm + nq_pad);
    const dim3 pg((unsigned)((nq_pad + 3) / 4));
    hipLaunchKernelGGL(vg_q8_query_prep_kernel, pg, dim3(256), 0, stream, dev_xqueries, xstride, dim, nq_real, nq_pad, mode, (const int *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, qkeys, qscales, (uint8_t *)nullptr, (uint8_t *)nullptr, q8stride, (float4 *)nullptr, type_code);
    hipLaunchKernelGGL(vg_q8_rank_kernel, dim3((unsigned)(nq_pad / 4)), dim3(256), 0, stream, (const float *)qkeys, (const float *)qscales, nq_pad, perm, common);
    hipLaunchKernelGGL(vg_q8_query_prep_kernel, pg, dim3(256), 0, stream, dev_xqueries, xstride, dim, nq_real, nq_pad, mode, (const int *)perm,
                       (const float *)common, (const float *)qscales, (float *)nullptr, (float *)nullptr, xq_sorted, qcodes, q8stride, qstat, type_code);
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    const int flag_index = vg_batch_q8_regions(nq_pad, npart);
    // k-steps of the exact kernel's shape: an f32 corpus is described by its bf16 image (what picks the chunks per lane), f16 / bf16 by themselves
    const int xntb = type_code == 2 ? (int)((((long long)dim * 2 + 15) / 16 * 16 + 31) / 32) : (int)((xstride + 31) / 32);
    const size_t smem_exact = (size_t)VGH_QPW * (8 + 4 + 4 + 4) + (size_t)VGH_QPW * k * 8 + (size_t)pair_cap * 8;      // (+ the region's pairs)
    if ((rc = (int)hipMemsetAsync(dev_pair_counts + flag_index, 0, sizeof(uint32_t), stream)) != 0) return rc;

    BatchArgsH hx;                                                    // the exact-evaluation kernel's view (vg_batch_hx_kernel)
    hx.rows = nullptr; hx.tiled = 1; hx.queries = xq_sorted; hx.xrows = dev_xrows; hx.xqueries = xq_sorted; hx.xstride = xstride;
    hx.cerr = 0.0f; hx.row_nn = dev_xnorm; hx.cand = dev_cand; hx.n_rows = n_rows; hx.stride = q8stride;
    hx.nq_pad = nq_pad; hx.nq_real = nq_pad; hx.k = k; hx.mode = mode; hx.root = root; hx.dim = dim;      // (sorted slots: padding and unjudged queries sit at the end - the filter passes none of their pairs)
    hx.pairs = dev_pairs; hx.pair_counts = dev_pair_counts; hx.pair_cap = pair_cap; hx.qnn = nullptr; hx.evals = dev_evals; hx.lds_pairs = 1; hx.part_group = 0;
    hx.tiles_per_part = 0; hx.tile_begin = 0; hx.tile_end = 0; hx.part_base = 0;

    BatchArgsQ8 a;
    a.rows = dev_rows_tm; a.rstat = reinterpret_cast<const float4 *>(dev_rstat); a.qcodes = qcodes; a.qstat = qstat;
    a.n_rows = n_rows; a.stride = q8stride; a.nq_pad = nq_pad; a.k = k; a.mode = mode; a.root = root;
    a.rel = (float)(dim + 64) * 2.384185791015625e-7f;               // (D + 64) 2^-22
    a.part_base = 0;
    a.pairs = dev_pairs; a.pair_counts = dev_pair_counts; a.pair_cap = pair_cap; a.flag_index = flag_index;
    // short rows: the wide form (512 queries per workgroup) whenever the padded batch is a multiple of 512 (VG_BATCH_H_WAVES=4: the 256-query form)
    const bool wide = !lcfg && nq_pad % 512 == 0 && vg_sw(SW_VG_BATCH_H_WAVES, 8) != 4;
    const bool narrow = !lcfg && nq_pad == 128;
    const size_t smem = lcfg ? vgql_lds_bytes(ntb) : (wide ? vgqw_lds_bytes(ntb) : (narrow ? vgqn_lds_bytes(ntb) : vgq_lds_bytes(ntb)));
    const bool long_exact = xstride > (type_code == 2 ? 4096 : 2048);       // rows beyond what the short exact kernel's chunks per lane cover
    const int G = narrow ? 1 : nq_pad / (wide ? 512 : VGQ_QPB);
    const int hx_waves = wide ? 16 : (narrow ? 4 : VGQ_WAVES * VGQ_QS);                    // 32-query regions per query group and partition
    // The batch in launches: (1) the bound-only PRE-PASS over the first `pre` tiles + the selection of every query's start threshold
    // (round 6; round 5 warmed the lists up with five tiny stages - two tiles with every gate open, then x8 x8 x8 x8 - that cost a fifth of
    // the batch); (2) STAGES over growing row ranges, the first one [0, VGQ_FIRST_MULT x pre) from the pre-pass' thresholds with empty
    // lists, every later one from every query's exact k-th best over all rows so far (lists merged after every stage): x4 while a stage is
    // small (its launches are what it costs), x2 from 1/16 of the corpus on (fewer pairs for the exact evaluation: ~k ln(growth) rows per
    // query truly enter, several times that pass the int8 bound).
    float *premin = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(common) + 16);
    const long long pre = std::min<long long>(VGQ_PRE_TILES, std::max<long long>(64, ntiles / 8)) / 8 * 8;
    long long bounds[24];
    int nstages = 0;
    {
        const int sw_growth = vg_sw(SW_VG_BATCH_STAGES, 0);
        // (long rows: an exact evaluation reads up to 6 KB - tighter thresholds pay for a few more launches: x1.5; 20.8 -> 20.0 ms at 10M x 1536)
        // (the 128-slot form: x4 to the end - a stage's launches weigh more than the pairs of so few queries: 1.17 -> 1.14 ms at 16 queries)
        const int late_growth = sw_growth > 100 ? sw_growth : (lcfg ? 150 : ((!lcfg && nq_pad == 128) ? 400 : 200));
        bounds[0] = 0;
        long long b = pre * (lcfg ? 1 : VGQ_FIRST_MULT);
        while (b < ntiles && nstages + 2 < 24 && ntiles - b > b / 4) {      // (no sliver at the end)
            bounds[++nstages] = b;
            // (the 128-slot form: early stages x16 up to 32 queries, x8 beyond - 4 queries 0.98 -> 0.92 ms, 16: 1.07 -> 1.02, 128: 1.18 -> 1.17)
            b = (b < ntiles / 16) ? b * ((!lcfg && nq_pad == 128) ? (nq_real <= 32 ? 16 : 8) : 4) : b * late_growth / 100;
        }
        bounds[++nstages] = ntiles;
    }
    auto launch_filter = [&](int blocks, bool is_pre) -> int {
        if (lcfg) {
            if (ntb == 12) return launch_q8l_mode<12, 2>(a, blocks, smem, stream, is_pre);
            if (ks == 2) return launch_q8l_mode<16, 2>(a, blocks, smem, stream, is_pre);
            return launch_q8l_mode<16, 3>(a, blocks, smem, stream, is_pre);
        }
        if (wide) {
            if (ntb == 4) return launch_q8w_mode<4>(a, blocks, smem, stream, is_pre);
            if (ntb == 8) return launch_q8w_mode<8>(a, blocks, smem, stream, is_pre);
            if (ntb == 12) return launch_q8w_mode<12>(a, blocks, smem, stream, is_pre);
            return launch_q8w_mode<16>(a, blocks, smem, stream, is_pre);
        }
        if (narrow) {
            if (ntb == 4) return launch_q8n_mode<4>(a, blocks, smem, stream, is_pre);
            if (ntb == 8) return launch_q8n_mode<8>(a, blocks, smem, stream, is_pre);
            if (ntb == 12) return launch_q8n_mode<12>(a, blocks, smem, stream, is_pre);
            return launch_q8n_mode<16>(a, blocks, smem, stream, is_pre);
        }
        if (ntb == 4) return launch_q8_mode<4>(a, blocks, smem, stream, is_pre);
        if (ntb == 8) return launch_q8_mode<8>(a, blocks, smem, stream, is_pre);
        if (ntb == 12) return launch_q8_mode<12>(a, blocks, smem, stream, is_pre);
        return launch_q8_mode<16>(a, blocks, smem, stream, is_pre);
    };
    {   // (1) the pre-pass
        a.tile_begin = 0; a.tile_end = pre;
        const int np = (int)std::min<long long>(npart, std::max<long long>(8, (pre / 4) / 8 * 8));
        a.npart = np; a.npart_total = np;
        a.tiles_per_part = (int)((pre + np - 1) / np);
        a.init_keys = nullptr; a.premin = premin;
        if ((rc = launch_filter(G * np, true)) != 0) return rc;
        hipLaunchKernelGGL(vg_q8_pre_select_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, stream, (const float *)premin, (int)pre, nq_pad, k, dev_out_keys);
        if ((rc = (int)hipGetLastError()) != 0) return rc;
    }
    a.premin = nullptr;
    for (int s = 0; s < nstages; ++s) {
        a.tile_begin = bounds[s]; a.tile_end = bounds[s + 1];
        const long long tiles_s = a.tile_end - a.tile_begin;
        // a stage of few tiles runs over fewer partitions: fewer regions for the exact-evaluation kernel, fewer lists for the merge
        const int np = (int)std::min<long long>(npart, std::max<long long>(8, (tiles_s / 4) / 8 * 8));
        a.npart = np; a.npart_total = np;
        a.tiles_per_part = (int)((tiles_s + np - 1) / np);
        a.init_keys = dev_out_keys;                                         // stage 0: the pre-pass' thresholds (k-th key only), then the merged lists
        hx.npart = np; hx.npart_total = np; hx.n_regions = vg_batch_q8_regions(nq_pad, np);
        hx.init_keys = a.init_keys; hx.seed = (s > 0) ? 1 : 0;             // (stage 0 starts with empty lists: the thresholds stand for no list entries)
        // one exact-evaluation block per 32 queries and pg consecutive partitions: 16 .. 31 lists per query reach the merge instead of up to 128
        int pg = (np % 8 == 0 && np / 8 >= 16) ? 8 : ((np % 4 == 0 && np / 4 >= 16) ? 4 : ((np % 2 == 0 && np / 2 >= 16) ? 2 : 1));
        if (pg > ((narrow || np > 256) ? 8 : VGQ_HX_GROUP)) pg = (narrow || np > 256) ? 8 : VGQ_HX_GROUP;   // (the 128-slot form's 512 partitions: 64 lists per query; its walks are short)
        if (s == 0 && VGQ_HX_GROUP_FIRST) pg = 1;                           // (the first stage has several hundred pairs per query: its walks are long enough)
        hx.part_group = pg;
        const int hx_blocks = hx.n_regions / pg, lists = np / pg;
        if ((rc = launch_filter(G * np, false)) != 0) return rc;
        if (long_exact)
            rc = type_code == 2 ? vghl_exact_regions_f32(&hx, hx_waves, hx_blocks, smem_exact, stream)
                                : (type_code == 1 ? vghl_exact_regions_bf16(&hx, hx_waves, hx_blocks, smem_exact, stream)
                                                  : vghl_exact_regions_f16(&hx, hx_waves, hx_blocks, smem_exact, stream));
        else
        rc = type_code == 2 ? vgh_launch_exact_f32(&hx, xntb, hx_waves, hx_blocks, smem_exact, stream)
                            : (type_code == 1 ? vgh_launch_exact_bf16(&hx, xntb, hx_waves, hx_blocks, smem_exact, stream)
                                              : vgh_launch_exact_f16(&hx, xntb, hx_waves, hx_blocks, smem_exact, stream));
        if (rc != 0) return rc;
        if ((rc = vg_batch_merge_launch(dev_cand, nq_pad, lists, lists, k, dev_out_keys, stream)) != 0) return rc;
    }
    return 0;
}
