// vg_batch_h_defs.h - what the half-precision batch kernels share: vg_batch_h.hip (rows up to 2 KiB: the A operand of a wavefront's 32 queries
// in its registers) and vg_batch_hl.hip (longer rows: the K dimension split over the wavefronts of a workgroup) - the launch arguments,
// the matrix-core wrapper, and the exact-evaluation kernel of the split form (filter kernel -> candidate pairs -> this).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vg_accum.h"
#include "vg_batch_common.h"

typedef _Float16 vgh_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 vgh_bf16x8 __attribute__((ext_vector_type(8)));
typedef float vgh_f32x16 __attribute__((ext_vector_type(16)));
typedef int vgh_i32x4 __attribute__((ext_vector_type(4)));

#define VGH_QPW 32
#define VGH_TILE 32
#define VGH_MAX_K 32
#define VGH_NORM_LO 1.0e-30f            // sum x^2 outside [LO, HI] (or NaN): the filter does not judge the row / query
#define VGH_NORM_HI 1.0e30f
#define VGH_ACCEPT 3.0e38f
static __device__ __attribute__((aligned(16))) uint32_t vgh_zero_chunk[4] = {0u, 0u, 0u, 0u};   // where a lane with nothing to load points its load

enum { VGH_DOT = 0, VGH_COS = 1, VGH_L2 = 2 };
enum { VGH_REAL = 0, VGH_BOUNDK = 1, VGH_FILTER = 2 };

struct BatchArgsH {
    const uint8_t *rows;      // N x stride bytes (f16 / bf16 elements, zero padded to 16 bytes): what the matrix core reads -
                              // row-major, or (tiled != 0) the TILE-MAJOR copy of vg_batch_i8.hip: tile t = rows 32t .. 32t+31 =
                              // 32 * stride contiguous bytes, chunk column c of the 32 rows at c * 512 + row * 16
    int tiled;
    const uint8_t *queries;   // nq_pad x stride bytes, zero padded (f32 corpora: unused, the A operand is converted from xqueries)
    const uint8_t *xrows;     // what the exact evaluation reads: = rows, or the f32 corpus behind a bf16 shadow copy
    const uint8_t *xqueries;  // = queries, or the f32 queries (nq_pad x xstride bytes, zero padded)
    long long xstride;
    float cerr;               // relative error bound of the filter's s~ (times |q||x|)
    const float *row_nn;      // (float) sum x^2 per row - f32 corpora: ||x|| - readable for four tiles past the last row
    uint64_t *cand;
    long long n_rows;
    long long stride;
    int nq_pad, nq_real, npart, k;
    int mode, root, dim;
    int tiles_per_part;
    long long tile_begin, tile_end;
    int part_base, npart_total;
    const uint64_t *init_keys;
    int seed;                 // staged real passes (vg_batch_common.h): partition 0 starts its lists from init_keys
    unsigned long long *evals;   // exact evaluations of the real passes (one atomic per wavefront; the host's selectivity guard), or NULL
    // ---- the split form (KIND = FILTER + vg_batch_hx_kernel): every wavefront of the filter kernel owns one REGION of pair_cap pairs,
    // region = ((g * npart_total + part_base + part) * waves + wave); a pair = (query in the wavefront's 32) << 32 | row, appended in
    // scan order; pair_counts[region] = pairs written, pair_counts[n_regions] = overflow flag (a region was full: the host repeats
    // the batch through the fused kernel)
    uint64_t *pairs;
    uint32_t *pair_counts;
    int pair_cap, n_regions;
    const float *qnn;         // vg_batch_hl.hip: (float) sum q^2 per query (nq_pad), made by the host once per batch
    int part_group;           // vg_batch_hx_kernel (round 6): > 1 = ONE block walks the regions of part_group consecutive partitions of its 32 queries, one after the
                              // other in scan order, into ONE set of lists: a late stage of the int8 filter has ~17 pairs per region, and 4 096 blocks that each
                              // load 32 thresholds and write 32 x 64 keys (16 KB, which the merge reads again) cost ~65 us whatever the pairs; npart_total must be
                              // a multiple, part_base 0, subs 1; the lists come out as npart_total / part_group lists per query
    int lds_pairs;            // vg_batch_hx_kernel: copy a region's pairs into LDS first (the launch provides pair_cap * 8 more bytes of it): the walk
                              // over them - 64 pair words per look, one look per pair - then costs LDS reads instead of an L2 round trip each
};

template <int CTRL> __device__ __forceinline__ uint64_t vgh_dpp64(uint64_t v) {
    return ((uint64_t)vg_dpp_u32<CTRL>((uint32_t)(v >> 32)) << 32) | vg_dpp_u32<CTRL>((uint32_t)v);
}
__device__ __forceinline__ uint64_t vgh_min64(uint64_t a, uint64_t b) { return a < b ? a : b; }

template <int VT>
__device__ __forceinline__ vgh_f32x16 vgh_mfma(const vgh_i32x4 &a, const vgh_i32x4 &b, const vgh_f32x16 &c) {
    if constexpr (VT == T_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vgh_f16x8, a), __builtin_bit_cast(vgh_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vgh_bf16x8, a), __builtin_bit_cast(vgh_bf16x8, b), c, 0, 0, 0);
}

// ---- the second half of the split form: the pairs the FILTER kernel let through, evaluated exactly.  One wavefront per region (the
// filter wavefront that wrote it: 32 queries, one partition), its pairs in scan order - the same arithmetic (Accum of the single-query
// kernel, 64 lanes per row), the same strict insertion into the same 32 sorted lists in LDS, the same output layout as the fused
// kernel's slow path, so the lists are what that kernel would have written.  Small wavefronts (no A operand, no accumulators): a CU
// holds dozens of them and their row fetches overlap instead of stalling a streaming workgroup one at a time.
// subs: pair regions per block (1: the filter wavefront's own; vg_batch_hl.hip: the 2 or 4 wavefronts that finish the scores of one
// set of 32 queries write a region each - regions block * subs .. + subs - 1, read one after the other: a query's pairs all sit in ONE
// of them, in scan order).
// EIGHT wavefronts per block (four: + 1 .. 2 % per batch, profiles/r8q): wavefront v evaluates the pairs of the queries with (query & 7) == v - every wavefront reads all pair
// words of the block's regions and skips the others'.  A query's pairs stay with one wavefront, in order; its list, threshold and
// statistics in LDS are touched by that wavefront only.
#ifndef VGHX_WAVES
#define VGHX_WAVES 8
#endif
template <int VT, int MODE, int XU>
__global__ __launch_bounds__(64 * VGHX_WAVES) void vg_batch_hx_kernel(BatchArgsH a, int waves, int subs) {
    constexpr bool COS = (MODE == VGH_COS), L2M = (MODE == VGH_L2), XF32 = (VT == T_F32);
    constexpr int ACC = COS ? (XF32 ? A_COS : A_COSN) : (L2M ? A_L2 : A_DOT);
    typedef Accum<VT, ACC> Exact;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    double *qq_w = reinterpret_cast<double *>(smem);                     // [32] sum q^2
    uint32_t *qsp_w = reinterpret_cast<uint32_t *>(qq_w + VGH_QPW);      // [32] query holds Inf / NaN
    uint32_t *qhave = qsp_w + VGH_QPW;                                   // [32] the two above are valid
    float *thr_w = reinterpret_cast<float *>(qhave + VGH_QPW);           // [32] k-th best so far
    uint64_t *wave_lists = reinterpret_cast<uint64_t *>(thr_w + VGH_QPW);   // [32][k]
    const int lane = threadIdx.x & 63, k = a.k;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int PG = a.part_group > 1 ? a.part_group : 1;
    const int lists_total = a.npart_total / PG;                        // lists per query this launch writes
    const int wave = (int)(blockIdx.x % waves);
    const long long gp = blockIdx.x / waves;
    const int list_idx = (int)(gp % lists_total);                        // (PG == 1: the partition, counted from part_base)
    const int g = (int)(gp / lists_total);
    const int part = PG > 1 ? list_idx * PG : list_idx - a.part_base;
    if (part < 0 || part >= a.npart) return;
    const long long region = PG > 1 ? ((long long)(g * a.npart_total + part) * waves + wave) : (long long)blockIdx.x;
    const int q0 = g * (waves * VGH_QPW) + wave * VGH_QPW;
    const int xchunks = (int)(a.xstride / 16);
    if (threadIdx.x < VGH_QPW) {
        const int qi = threadIdx.x;
        float t = a.init_keys ? vgb_kth_distance(a.init_keys[(long long)(q0 + qi) * 64 + (k - 1)]) : INFINITY;
        if (q0 + qi >= a.nq_real) t = -INFINITY;
        thr_w[qi] = t;
        qhave[qi] = 0u;
    }
    {
        const bool seeded = a.seed != 0 && part == 0;
        for (int s = threadIdx.x; s < VGH_QPW * k; s += 64 * VGHX_WAVES)
            wave_lists[s] = seeded ? a.init_keys[(long long)(q0 + s / k) * 64 + s % k] : VG_EMPTY_KEY;
    }
    __syncthreads();
    unsigned n_all = 0;
    const int nsub = PG > 1 ? min(PG, a.npart - part) : subs;
    for (int sub = 0; sub < nsub; ++sub) {
    const long long reg_s = PG > 1 ? region + (long long)sub * waves : region * subs + sub;
    const unsigned n = a.pair_counts[reg_s];
    const uint64_t *my_pairs = a.pairs + reg_s * a.pair_cap;
    if (a.lds_pairs) {                                                // (block-uniform)
        uint64_t *lp = wave_lists + VGH_QPW * k;
        __syncthreads();                                              // (the previous sub-region's copy is no longer read)
        for (unsigned i = threadIdx.x; i < n; i += 64 * VGHX_WAVES) lp[i] = my_pairs[i];
        __syncthreads();
        my_pairs = lp;
    }
    n_all += n;
    // Two pair slots, A and B: the row (and query) of pair i+1 is in flight while pair i is evaluated - one wavefront walks its pairs in
    // order (a query's list insertions are ordered), and with one HBM round trip per pair exposed the kernel took half as long as the
    // filter kernel that feeds it (long rows, 6 KiB per evaluation: profiles/r7b_*)
    auto issue = [&](uint64_t pr, uint4 (&qv)[XU], uint4 (&xv)[XU], float &nn) __attribute__((always_inline)) {
        const int qi_u = __builtin_amdgcn_readfirstlane((int)(pr >> 32));
        const uint32_t row_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pr);
        const uint8_t *qp = a.xqueries + (long long)(q0 + qi_u) * a.xstride;
        const uint8_t *xp = a.xrows + (unsigned long long)row_u * (unsigned long long)a.xstride;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            qv[u] = make_uint4(0u, 0u, 0u, 0u); xv[u] = make_uint4(0u, 0u, 0u, 0u);
            if (lane + 64 * u < xchunks) { qv[u] = reinterpret_cast<const uint4 *>(qp)[lane + 64 * u]; xv[u] = reinterpret_cast<const uint4 *>(xp)[lane + 64 * u]; }
        }
        nn = a.row_nn[row_u];
    };
    auto evaluate = [&](uint64_t pr, const uint4 (&qv)[XU], const uint4 (&xv)[XU], float nn_u) __attribute__((always_inline)) {
        const int qi_u = __builtin_amdgcn_readfirstlane((int)(pr >> 32));
        const uint32_t row_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pr);
        const uint8_t *qp = a.xqueries + (long long)(q0 + qi_u) * a.xstride;
        const uint8_t *xp = a.xrows + (unsigned long long)row_u * (unsigned long long)a.xstride;
        if (qhave[qi_u] == 0u) {                                         // the query's statistics, on first use (wave-uniform branch)
            if constexpr (XF32) {
                const typename Accum<T_F32, A_COS>::QStat st = Accum<T_F32, A_COS>::template query_stat<XU>(qv, 6);
                if (lane == 0) { qq_w[qi_u] = (double)st.qq; qsp_w[qi_u] = 0u; qhave[qi_u] = 1u; }
            } else {
                const typename Accum<VT, A_COSN>::QStat st = Accum<VT, A_COSN>::template query_stat<XU>(qv, 6);
                if (lane == 0) { qq_w[qi_u] = st.qq; qsp_w[qi_u] = st.qspecial; qhave[qi_u] = 1u; }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // (this wavefront's own LDS writes, read back below: in order)
            __builtin_amdgcn_wave_barrier();
        }
        typename Exact::QStat qs;
        if constexpr (XF32) qs.qq = (float)qq_w[qi_u];
        else { qs.qq = qq_w[qi_u]; qs.qspecial = qsp_w[qi_u]; }
        Exact acc;
        acc.init();
#pragma unroll
        for (int u = 0; u < XU; ++u) acc.chunk(qv[u], xv[u]);
        float d;
        if constexpr (XF32) {
            d = acc.finish(qs, 6, a.root);
        } else {
            if constexpr (COS) d = acc.finish_cached_norm(qs, 6, nn_u);
            else d = acc.finish(qs, 6, a.root);
            if (__builtin_amdgcn_readfirstlane((int)acc.special(qs, 6)) != 0)
                d = vg_slow_distance<VT, (COS ? A_COS : ACC)>(reinterpret_cast<const uint16_t *>(qp), reinterpret_cast<const uint16_t *>(xp), a.dim, a.root);
        }
        const float de = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, vg_clamp(d))));
        const float thr_u = thr_w[qi_u];
        if (!(de < thr_u)) return;                                       // strict: rows arrive in scan order (see the fused kernel)
        const float nt = vgb_kth_distance(vgb_list_insert(wave_lists + qi_u * k, k, lane, vg_make_key(de, row_u)));
        if (nt < thr_u && lane == 0) thr_w[qi_u] = nt;
    };
    // the next pair of THIS wavefront at or behind position `from` (64 pair words per look: lane l reads word from + l); n if there is none
    auto next_mine = [&](unsigned from, uint64_t &pr) __attribute__((always_inline)) -> unsigned {
        for (; from < n; from += 64) {
            const uint64_t w = (from + lane < n) ? my_pairs[from + lane] : 0ull;
            const unsigned long long mine = __ballot(from + lane < n && (int)((w >> 32) & (unsigned)(VGHX_WAVES - 1)) == wv);
            if (mine != 0ull) {
                const int src = __ffsll((long long)mine) - 1;
                pr = vg_readlane64(w, src);
                return from + (unsigned)src;
            }
        }
        return n;
    };
    uint4 qvA[XU], xvA[XU], qvB[XU], xvB[XU];
    float nnA = 0.0f, nnB = 0.0f;
    uint64_t prA = 0ull, prB = 0ull;
    unsigned iA = next_mine(0u, prA), iB = n;
    if (iA < n) { issue(prA, qvA, xvA, nnA); iB = next_mine(iA + 1u, prB); }
    while (iA < n) {
        if (iB < n) issue(prB, qvB, xvB, nnB);
        evaluate(prA, qvA, xvA, nnA);
        iA = (iB < n) ? next_mine(iB + 1u, prA) : n;
        if (iB < n) {
            if (iA < n) issue(prA, qvA, xvA, nnA);
            evaluate(prB, qvB, xvB, nnB);
            iB = (iA < n) ? next_mine(iA + 1u, prB) : n;
        }
    }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < VGH_QPW * 64; s += 64 * VGHX_WAVES) {
        const int qi = s >> 6, slot = s & 63;
        a.cand[((long long)(q0 + qi) * lists_total + list_idx) * 64 + slot] = (slot < k) ? wave_lists[qi * k + slot] : VG_EMPTY_KEY;
    }
    if (a.evals && threadIdx.x == 0 && n_all) atomicAdd(a.evals, (unsigned long long)n_all);
}
