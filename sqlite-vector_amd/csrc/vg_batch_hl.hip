// vg_batch_hl.hip - batched queries over LONG rows (1025 .. 3072 elements: f16 / bf16 corpora, f32 corpora through their bf16 shadow
// copy) on the matrix cores: the K dimension split over the wavefronts of a workgroup.
//
// vg_batch_h.hip keeps the A operand - a wavefront's 32 queries, whole rows - in registers: 4 registers per 32-byte k-step, 256 at
// 1024 elements, and that is where it ends (reference rows are often longer: 1536- and 3072-element embeddings).  Here a workgroup is
// FOUR wavefronts (one per SIMD, 512 registers each) that share ONE set of queries and split their rows four ways:
//   * wavefront p keeps k-steps [p * NTBP, (p + 1) * NTBP) of QS x 32 queries (QS = 2 sets for NTBP <= 24, else 1) - up to 256 registers;
//   * per 32-row tile it multiplies ITS slice of the tile (read straight from the tile-major copy into registers: each k-step is one
//     1 KiB-contiguous buffer_load_dwordx4 per wavefront; no LDS - no two wavefronts of the workgroup read the same bytes - the loads of
//     tile i + DEPTH are issued behind the MFMAs of tile i that free their registers) into partial scores, QS x 16 registers per lane;
//   * the four partial scores of every (query, row) pair meet in LDS: every wavefront writes its QS x 16 registers, one barrier - in the
//     middle of the NEXT tile's k loop, its LDS reads, sums and gate spread over the following k-steps - and wavefront w sums, and owns
//     from there on, registers [4 QS w, 4 QS (w + 1)) of the QS x 16: the gate of vg_batch_h.hip (same bounds: |s~ - s| <= c |q||x|,
//     c = (D + 64) 2^-21 [+ 2^-7 + 2^-16 behind a bf16 shadow]; rows of zeros are judged, NaN / Inf / out-of-range norms pass), then
//       FILTER kind: pairs that pass are appended to the wavefront's region of the pair buffer, evaluated exactly by
//                    vg_batch_hx_kernel (vg_batch_h_defs.h: the single-query kernel's arithmetic, strict insertion in scan order);
//       BOUNDK kind: the pre-pass - a pair enters its query's list with an UPPER BOUND of its distance, no exact evaluation.
// The workgroups of one partition (same rows, other queries) sit on one XCD (blockIdx & 7): the tile-major copy streams from HBM once
// per partition and from that XCD's L2 for the other query groups.
//
// Cost model: per tile a wavefront issues QS x NTBP MFMAs (32 cycles each) against NTBP KiB of loads - 64 queries per
// workgroup at 1536 elements means 96 KB per tile and CU through the vector L1 at 64 B / clk: 1536 cycles, the tile's own MFMA time.
// Measured: 0.36-0.38 of the bf16 peak at 1536 elements (MFMA pipe busy 44 %), 0.24 at 2048 / 3072 (32 queries per workgroup) - 70x
// the default single-query path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "vg_accum.h"
#include "vg_batch_common.h"
#include "vg_batch_h_defs.h"

#define VGHL_WAVES 4
#define VGHL_QS(NTBP) ((NTBP) <= 24 ? 2 : 1)     // (two sets at 32 k-steps = 256 registers of A: the gate spills, its reloads drain the load ring)
#define VGHL_MAX_KSTEPS (4 * 48)        // 6 KiB of half-precision elements per row: 3072
#ifndef VGHL_ABLATE
#define VGHL_ABLATE 0                   // measurement builds (wrong results): 1 = no meeting in LDS / gate, 2 = no loads in the tile loop, 3 = neither
#endif
#ifndef VGHL_MEET_AT
#define VGHL_MEET_AT(NTBP) ((NTBP) / 2)   // the k-step in front of which the previous tile's partial scores meet (barrier + LDS reads)
#endif
#ifndef VGHL_MEET_GAP
#define VGHL_MEET_GAP 6                 // k-steps of MFMAs between the LDS reads of the previous tile's partial scores and their use (3: + 2 %, profiles/r7s)
#endif
#ifndef VGHL_DEPTH
#define VGHL_DEPTH(NTBP) ((NTBP) <= 32 ? 2 : 1)     // tiles of B in flight per wavefront (registers: DEPTH x NTBP x 4 next to A's QS x NTBP x 4)
#endif

template <int VT, int NTBP, int MODE, int KIND>
__global__ __launch_bounds__(64 * VGHL_WAVES, 1) void vg_batch_hl_kernel(BatchArgsH a) {
    constexpr bool BOUND = (KIND == VGH_BOUNDK);
    constexpr int QS = VGHL_QS(NTBP), F = 4 * QS, NQ = QS * VGH_QPW;
    constexpr bool COS = (MODE == VGH_COS), L2M = (MODE == VGH_L2), XF32 = (VT == T_F32);
    constexpr int FT = XF32 ? T_BF16 : VT;                               // element type the matrix core multiplies
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float4 *red = reinterpret_cast<float4 *>(smem);                      // [2 buffers][4 wavefronts][QS * 4 register groups][64 lanes]
    float *qq_l = reinterpret_cast<float *>(red + 2 * VGHL_WAVES * QS * 4 * 64);   // [NQ] sum q^2
    float *thr_l = qq_l + NQ;                                            // [NQ] k-th best so far
    uint64_t *lists = reinterpret_cast<uint64_t *>(thr_l + NQ);          // [NQ][k]  (BOUNDK only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31, h = lane >> 5;
    const int k = a.k;
    const int G = a.nq_pad / NQ;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int g = idx % G;
    const int part = (idx / G) * 8 + xcd;
    if (part >= a.npart) return;
    const int q0 = g * NQ;
    const int chunks_per_row = (int)(a.stride / 16);
    const int kbase = wave * NTBP;                                       // this wavefront's first k-step

    // ---- per-query statistics: sum q^2, made once per batch by vg_query_norms_kernel (a loop over the queries here cost every workgroup
    // of every stage ~200 us: profiles/r7d_*); a query the filter cannot judge - Inf / NaN elements, a norm of zero or out of range -
    // carries the norm -1: it multiplies as zero, none of its pairs passes, vg_batch_api.hip answers it with a scan of its own
    if (tid < NQ) {                                      // thresholds: the pre-pass bound; padding queries - and the slots of queries that
        const float n2 = a.qnn[q0 + tid];                //   left the batch (norm < 0) - never accept
        qq_l[tid] = fmaxf(n2, 0.0f);
        float t = a.init_keys ? vgb_kth_distance(a.init_keys[(long long)(q0 + tid) * 64 + (k - 1)]) : INFINITY;
        if (q0 + tid >= a.nq_real || n2 < 0.0f) t = -INFINITY;
        thr_l[tid] = t;
    }
    if constexpr (BOUND) {
        for (int s = tid; s < NQ * k; s += 64 * VGHL_WAVES) lists[s] = VG_EMPTY_KEY;
    }
    __syncthreads();

    // ---- A operand: lane (x, h) keeps bytes [32 t + 16 h, +16) of query x of every set, for this wavefront's k-steps
    vgh_i32x4 areg[QS][NTBP];
    vgb_static_for<0, QS>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const float qqx = qq_l[s * VGH_QPW + x];
        const bool qzero = !(qqx >= VGH_NORM_LO && qqx <= VGH_NORM_HI);   // multiplies as ZERO: its pairs all take the exact path
        if constexpr (XF32) {
            const uint8_t *qrow = a.xqueries + (long long)(q0 + s * VGH_QPW + x) * a.xstride;
            auto bf = [](uint32_t lo, uint32_t hi) -> int {
                const uint32_t l = (lo + 0x7FFFu + ((lo >> 16) & 1u)) >> 16, u = (hi + 0x7FFFu + ((hi >> 16) & 1u)) & 0xFFFF0000u;
                return (int)(l | u);
            };
            vgb_static_for<0, NTBP>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                const long long off = ((long long)32 * (kbase + t) + 16 * h) * 2;      // the same 8 elements in the f32 row
                uint4 v0 = make_uint4(0u, 0u, 0u, 0u), v1 = make_uint4(0u, 0u, 0u, 0u);
                if (!qzero && off + 16 <= a.xstride) v0 = *reinterpret_cast<const uint4 *>(qrow + off);
                if (!qzero && off + 32 <= a.xstride) v1 = *reinterpret_cast<const uint4 *>(qrow + off + 16);
                areg[s][t] = vgh_i32x4{bf(v0.x, v0.y), bf(v0.z, v0.w), bf(v1.x, v1.y), bf(v1.z, v1.w)};
            });
        } else {
            const uint8_t *qrow = a.queries + (long long)(q0 + s * VGH_QPW + x) * a.stride;
            vgb_static_for<0, NTBP>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                const long long off = (long long)32 * (kbase + t) + 16 * h;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (!qzero && off < a.stride) v = *reinterpret_cast<const uint4 *>(qrow + off);
                areg[s][t] = vgh_i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            });
        }
    });

    // ---- the registers this wavefront finishes: flat f = F * wave + j (set f >> 4, register r = f & 15 of the 32 x 32 block:
    // query (r & 3) + 8 (r >> 2) + 4 h of the set, row x), all of one set
    const int fbase = F * wave, my_set = fbase >> 4, rbase = fbase & 15;
    const int qset0 = my_set * VGH_QPW;                                  // first query (within the workgroup's NQ) of that set
    const float cerr = a.cerr;
    const bool l2_root = a.root != 0;
    float init_f[F], gmul_f[F];
    uint32_t live = 0u;                                  // bit j: register j's query is a real one (not padding, not taken out of the batch)
    auto set_gate = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int r = rbase + j, qi = qset0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float thr = thr_l[qi], qqf = qq_l[qi], na = sqrtf(qqf);
        const bool qforce = !(qqf >= VGH_NORM_LO && qqf <= VGH_NORM_HI);
        const bool open = qforce || !(thr < VGH_ACCEPT);                 // +Inf / NaN threshold: accept everything
        if (COS) {
            const float Gf = (1.0f - thr) * na;
            init_f[j] = 1e-30f;
            gmul_f[j] = open ? VGH_ACCEPT : -(Gf - 1e-5f * fabsf(Gf) - cerr * na);
        } else if (L2M) {
            const float thr2 = l2_root ? thr * thr : thr;
            const float v = 0.5f * (thr2 * (1.0f + 1e-5f) - (1.0f - cerr) * qqf) + 1e-30f;
            init_f[j] = (open || !(v < VGH_ACCEPT)) ? VGH_ACCEPT : v;
            gmul_f[j] = -1.0f;
        } else {
            init_f[j] = open ? VGH_ACCEPT : thr + 1e-5f * fabsf(thr) + 1e-30f;
            gmul_f[j] = qforce ? 0.0f : cerr * na;
        }
        if (BOUND && open) {                             // the bound pass turns s~ into a bound: "accept everything" is a huge multiplier
            init_f[j] = COS ? 1e-30f : 0.0f;
            gmul_f[j] = VGH_ACCEPT;
        }
        if (thr == -INFINITY) {                          // padding queries never pass
            init_f[j] = COS ? 0.0f : -VGH_ACCEPT;
            gmul_f[j] = COS ? -VGH_ACCEPT : (L2M ? -1.0f : 0.0f);
            live &= ~(1u << j);
        } else live |= 1u << j;
    };
    vgb_static_for<0, F>([&](auto jc) __attribute__((always_inline)) { set_gate(jc); });

    const long long tile_first = a.tile_begin + (long long)part * a.tiles_per_part;
    const long long tile_last = min(tile_first + a.tiles_per_part, a.tile_end);
    const unsigned long long tile_bytes = (unsigned long long)VGH_TILE * (unsigned long long)a.stride;
    // k-step kbase + t of tile T: chunk columns 2 (kbase + t) + h of row x = bytes [(kbase + t) * 1024 + lane * 16, +16) of the tile.
    // Read through a BUFFER resource that spans exactly the tile: one wave-uniform base per tile, a 32-bit lane offset, the k-step in the
    // scalar offset - no per-lane 64-bit address per load - and a column past the row's last chunk (an odd chunk count, k-steps past the
    // row) lies past the tile's end, where a buffer load returns ZEROS instead of the next tile's bytes (0 x NaN would be NaN).
    auto tile_rsrc = [&](long long tile) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.rows) + (unsigned long long)tile * tile_bytes, 0, (int)tile_bytes, 0x00020000);
    };
    const int lane_off = lane * 16;
    auto load_b = [&](__amdgpu_buffer_rsrc_t rs, auto tc) __attribute__((always_inline)) -> vgh_i32x4 {
        constexpr int t = decltype(tc)::value;
        return __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, (kbase + t) * 1024, 0);
    };

    // FILTER kind: this wavefront's region of the pair buffer (4 per workgroup and partition; vg_batch_hx_kernel reads the 4 / QS regions
    // of a set one after the other)
    const long long region = ((long long)(g * a.npart_total + a.part_base + part)) * VGHL_WAVES + wave;
    uint64_t *my_pairs = BOUND ? nullptr : a.pairs + region * a.pair_cap;
    unsigned n_pairs = 0;                                                // (wave-uniform)

    // B ring: DEPTH tiles of this wavefront's k-steps in registers.  A load is issued right behind the MFMAs that free its registers, DEPTH
    // tiles ahead: what is in flight per CU (4 wavefronts x DEPTH x NTBP KiB: 192 KiB) is what bounds the stream - with one tile
    // (96 KiB at 24 k-steps) the first version ran at ~19 B / clk / CU from L2, 30 % of the matrix rate (profiles/r7b_*)
    constexpr int DEPTH = BOUND ? 1 : VGHL_DEPTH(NTBP);                  // (the pre-pass - 1 / 512 of the rows - keeps its registers for the lists' code)
    vgh_i32x4 breg[DEPTH][NTBP];
    float nn_ring[DEPTH];
    if (tile_first < tile_last) {
        vgb_static_for<0, DEPTH>([&](auto dc) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value;
            const long long tj = min(tile_first + d, tile_last - 1);
            const __amdgpu_buffer_rsrc_t rs0 = tile_rsrc(tj);
            nn_ring[d] = a.row_nn[tj * VGH_TILE + x];                    // (the order of the tile loop: norms, then the tile's k-steps)
            vgb_static_for<0, NTBP>([&](auto tc) __attribute__((always_inline)) { breg[d][decltype(tc)::value] = load_b(rs0, tc); });
        });
    }
    // One tile = its k loop, then the four partial scores MEET IN LDS: every wavefront writes its QS x 16 registers, one barrier, and
    // wavefront w sums registers [F w, F w + F).  The meeting of tile i is finished in the MIDDLE of tile i+1's k loop (finish_prev):
    // a wavefront that is early at the barrier has issued half of the next tile's MFMAs by then instead of idling from the end of its k
    // loop on - one wavefront per SIMD, nobody else feeds its matrix pipe (ablation, 1024 x 2M x 1536: MFMA alone 4.3 ms, + loads 6.0,
    // + meeting / gate / pairs at the tile's end 8.4, + exact evaluation 9.5: profiles/r7f_*).  Two LDS buffers: tile i+2 is written
    // behind barrier i+1, which every wavefront reaches with its reads of tile i done.
    bool have_prev = false;
    float nn_prev = 0.0f;
    long long tile_prev = 0;                                           // (wave-uniform; the row is made from it and the lane where it is used)
    int par_prev = 0;
    // (in two halves: the barrier and the LDS reads - and, VGHL_MEET_GAP k-steps of MFMAs later, when they have landed, the sums, the gate
    // and the pairs)
    float4 met[QS][VGHL_WAVES];
    auto meet_prev = [&]() __attribute__((always_inline)) {
        __syncthreads();
        const float4 *red_r = red + par_prev * VGHL_WAVES * (QS * 4 * 64);
#pragma unroll
        for (int j4 = 0; j4 < QS; ++j4)
#pragma unroll
            for (int src = 0; src < VGHL_WAVES; ++src) met[j4][src] = red_r[(src * QS * 4 + wave * QS + j4) * 64 + lane];
    };
    // (... then one register group's sums per k-step - pairs of floats as they lie in the registers: v_pk_add_f32 without shuffles - and
    // the gate: a burst of ~100 VALU instructions between two MFMAs idles the matrix pipe of a SIMD with one wavefront, spread over the
    // k-steps they run in the MFMAs' shadow)
    typedef float vghl_f2 __attribute__((ext_vector_type(2)));
    float fin[F];
    auto sum_prev = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j4 = decltype(jc)::value;
        const vghl_f2 lo = (vghl_f2{met[j4][0].x, met[j4][0].y} + vghl_f2{met[j4][1].x, met[j4][1].y}) + (vghl_f2{met[j4][2].x, met[j4][2].y} + vghl_f2{met[j4][3].x, met[j4][3].y});
        const vghl_f2 hi = (vghl_f2{met[j4][0].z, met[j4][0].w} + vghl_f2{met[j4][1].z, met[j4][1].w}) + (vghl_f2{met[j4][2].z, met[j4][2].w} + vghl_f2{met[j4][3].z, met[j4][3].w});
        fin[4 * j4] = lo.x; fin[4 * j4 + 1] = lo.y; fin[4 * j4 + 2] = hi.x; fin[4 * j4 + 3] = hi.y;
    };
    auto finish_prev = [&]() __attribute__((always_inline)) {
        uint32_t x_now;                                                  // lane & 31, made HERE: as a loop invariant it is spilled, and its reload drains the load ring
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_and_b32 %0, 31, %0" : "=v"(x_now));
        const long long row_prev = tile_prev * VGH_TILE + x_now;
        // ---- the gate (vg_batch_h.hip): fin + init + gmul * lane_term >= 0
        // A row of ZEROS is judged like any other (every product is exactly 0, the bound's |x| term too) - only NaN / Inf / out-of-range
        // norms send a row's pairs down the exact path whatever their scores: a corpus with a fraction of empty vectors would
        // otherwise fill the pair regions with them.  Cosine is the exception in form only: the reference gives a zero-norm row the
        // distance 1.0 (distance-cpu.c:74-110), so its pairs pass iff the query's threshold has not dropped below that.
        const bool zero_row = (nn_prev == 0.0f);
        const bool force = !(nn_prev <= VGH_NORM_HI) || (nn_prev < VGH_NORM_LO && !zero_row);      // NaN / Inf / out of range
        const float lane_term = force ? 0.0f : (L2M ? 0.5f * (1.0f - cerr) * nn_prev : sqrtf(nn_prev));
        uint32_t mybits = 0u;
        vgb_static_for<0, F>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const bool pass = force || fmaf(gmul_f[j], lane_term, fin[j] + init_f[j]) >= 0.0f;
            mybits |= pass ? (1u << j) : 0u;
        });
        if (COS && __ballot(zero_row) != 0ull) {         // (rare: the thresholds come from LDS)
            uint32_t keep = 0u;
            vgb_static_for<0, F>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const int r = rbase + j;
                keep |= (thr_l[qset0 + (r & 3) + 8 * (r >> 2) + 4 * h] >= 0.99999f) ? (1u << j) : 0u;
            });
            if (zero_row) mybits &= keep;
        }
        mybits &= live;                                  // (also under a row that sends every pair down the exact path)
        if (!(row_prev < a.n_rows)) mybits = 0u;
        if (__ballot(mybits != 0u) != 0ull) {
            if constexpr (BOUND) {
                bool changed = false;
                vgb_static_for<0, F>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    if (__ballot((mybits >> j) & 1u) == 0ull) return;
                    // upper bound of the distance from the filter's estimate s~
                    const int r = rbase + j, q_lo = qset0 + (r & 3) + 8 * (r >> 2), qi_lane = q_lo + 4 * h;
                    const float qqf = qq_l[qi_lane], na = sqrtf(qqf), nb = sqrtf(nn_prev);
                    const float st = fin[j], E = cerr * na * nb;
                    float ub;
                    if (COS) ub = fminf(1.0f - st / (na * nb) + cerr + 1e-5f, 2.0f);
                    else if (L2M) { const float d2 = fmaxf(qqf + nn_prev - 2.0f * st + 2.0f * E + 1e-5f * (qqf + nn_prev), 0.0f); ub = l2_root ? sqrtf(d2) : d2; }
                    else ub = -st + E;
                    ub = ub + 1e-5f * fabsf(ub) + 1e-30f;
                    const bool qforce = !(qqf >= VGH_NORM_LO && qqf <= VGH_NORM_HI);
                    const bool ok = ((mybits >> j) & 1u) && (q0 + qi_lane < a.nq_real) && !force && !qforce && (ub < thr_l[qi_lane]);
                    if (__ballot(ok) == 0ull) return;
                    // only the SMALLEST bound of each query in this tile enters its list (k entries then stand for k different rows all
                    // the same, and a list costs one insert per (query, tile))
                    uint64_t key = ok ? vg_make_key(ub, (uint32_t)row_prev) : VG_EMPTY_KEY;
                    key = vgh_min64(key, vgh_dpp64<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(key));
                    key = vgh_min64(key, vgh_dpp64<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(key));
                    key = vgh_min64(key, vgh_dpp64<VG_DPP_ROW_HALF_MIRROR>(key));
                    key = vgh_min64(key, vgh_dpp64<VG_DPP_ROW_MIRROR>(key));
                    key = vgh_min64(key, (uint64_t)__shfl_xor((unsigned long long)key, 16));       // lanes 0-31 / 32-63: one query each
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint64_t c = vg_readlane64(key, 32 * hh);
                        if (c == VG_EMPTY_KEY) continue;
                        const int qi_u = q_lo + 4 * hh;
                        const float nt = vgb_kth_distance(vgb_list_insert(lists + qi_u * k, k, lane, c));
                        if (lane == 0) thr_l[qi_u] = nt;                 // (a list's k-th bound never grows; only this wavefront reads it)
                    }
                    changed = true;
                });
                if (changed) vgb_static_for<0, F>([&](auto jc) __attribute__((always_inline)) { set_gate(jc); });
            } else {
                // the passing pairs, lanes ascending (a query's rows ascending: scan order)
                unsigned long long any;
                while ((any = __ballot(mybits != 0u)) != 0ull) {
                    const int src = __ffsll((long long)any) - 1;
                    const uint32_t bits_u = (uint32_t)__builtin_amdgcn_readlane((int)mybits, src);
                    const int j_u = __ffs((int)bits_u) - 1;
                    if (lane == src) mybits &= mybits - 1u;
                    const int r_u = rbase + j_u, qi_u = (r_u & 3) + 8 * (r_u >> 2) + 4 * (src >> 5);   // within the set
                    if (q0 + qset0 + qi_u >= a.nq_real) continue;
                    const uint32_t row_u = (uint32_t)__builtin_amdgcn_readlane((int)row_prev, src);
                    if (n_pairs < (unsigned)a.pair_cap) { if (lane == 0) my_pairs[n_pairs] = ((uint64_t)(uint32_t)qi_u << 32) | row_u; }
                    else if (lane == 0) a.pair_counts[a.n_regions] = 1u;     // region full: the host answers this batch another way
                    ++n_pairs;
                }
            }
        }
    };
    auto do_tile = [&](long long tile, auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        constexpr int P = VGHL_MEET_AT(NTBP);                            // k-steps in front of the previous tile's meeting
        const long long ti = tile - tile_first;
        const long long tile_next = min(tile + DEPTH, tile_last - 1);
        const __amdgpu_buffer_rsrc_t rs_next = tile_rsrc(tile_next);
        float nn_row = nn_ring[d];
        nn_ring[d] = a.row_nn[tile_next * VGH_TILE + x];                 // (in front of that tile's loads: it lands first)
        vgh_f32x16 acc[QS];
#pragma unroll
        for (int s = 0; s < QS; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
        auto k_step = [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            const vgh_i32x4 b = breg[d][t];
            vgb_static_for<0, QS>([&](auto sc) __attribute__((always_inline)) { constexpr int s = decltype(sc)::value; acc[s] = vgh_mfma<FT>(areg[s][t], b, acc[s]); });
            if constexpr (!(VGHL_ABLATE & 2)) breg[d][t] = load_b(rs_next, tc);     // the same k-step of the tile DEPTH ahead
        };
        constexpr int P2 = P + VGHL_MEET_GAP;
        static_assert(P2 + QS < NTBP, "the meeting's steps must lie inside the k loop");
        vgb_static_for<0, NTBP>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            if constexpr ((VGHL_ABLATE & 1) == 0) {
                if constexpr (t == P) { if (have_prev) meet_prev(); }
                if constexpr (t >= P2 && t < P2 + QS) { if (have_prev) sum_prev(std::integral_constant<int, t - P2>{}); }
                if constexpr (t == P2 + QS) { if (have_prev) finish_prev(); }
            }
            k_step(tc);
        });
        if constexpr ((VGHL_ABLATE & 1) != 0) {
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[0][15]), "v"(acc[QS - 1][0]), "v"(acc[QS - 1][15]));     // (keep the MFMA chains alive)
            return;
        }
        if constexpr (XF32) nn_row = nn_row * nn_row;                    // (the f32 corpus caches ||x||, not sum x^2)
        float4 *red_w = red + ((int)(ti & 1) * VGHL_WAVES + wave) * (QS * 4 * 64);
        vgb_static_for<0, QS>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                red_w[(s * 4 + q4) * 64 + lane] = make_float4(acc[s][4 * q4], acc[s][4 * q4 + 1], acc[s][4 * q4 + 2], acc[s][4 * q4 + 3]);
        });
        have_prev = true; nn_prev = nn_row; tile_prev = tile; par_prev = (int)(ti & 1);
    };
    for (long long tile0 = tile_first; tile0 < tile_last; tile0 += DEPTH) {
        vgb_static_for<0, DEPTH>([&](auto dc) __attribute__((always_inline)) {
            if (tile0 + decltype(dc)::value < tile_last) do_tile(tile0 + decltype(dc)::value, dc);     // (wave- and workgroup-uniform)
        });
    }
    if constexpr ((VGHL_ABLATE & 1) == 0) {
        if (have_prev) { meet_prev(); vgb_static_for<0, QS>([&](auto jc) __attribute__((always_inline)) { sum_prev(jc); }); finish_prev(); }
    }
    if constexpr (!BOUND) {
        if (lane == 0) a.pair_counts[region] = n_pairs < (unsigned)a.pair_cap ? n_pairs : (unsigned)a.pair_cap;
    } else {
        __syncthreads();
        for (int s = tid; s < NQ * 64; s += 64 * VGHL_WAVES) {
            const int qi = s >> 6, slot = s & 63;
            a.cand[((long long)(q0 + qi) * a.npart_total + a.part_base + part) * 64 + slot] = (slot < k) ? lists[qi * k + slot] : VG_EMPTY_KEY;
        }
    }
}

// ---- host side: six units (build.py: -DVGHL_TU=0 .. 5, element type x kind; unit 0 holds the entry point)
#ifndef VGHL_TU
#define VGHL_TU 0
#endif

static int vghl_ntbp(long long stride_bytes) {
    const int ksteps = (int)((stride_bytes + 31) / 32);
    if (ksteps <= 64 || ksteps > VGHL_MAX_KSTEPS) return 0;              // (up to 64: vg_batch_h.hip)
    const int per = (ksteps + VGHL_WAVES - 1) / VGHL_WAVES;
    return per <= 24 ? 24 : (per <= 32 ? 32 : 48);
}
static size_t vghl_lds_bytes(int ntbp, int k, bool bound) {
    const int qs = VGHL_QS(ntbp);
    return (size_t)2 * VGHL_WAVES * qs * 4 * 64 * 16 + (size_t)qs * VGH_QPW * 8 + (bound ? (size_t)qs * VGH_QPW * k * 8 : 0);
}
template <int VT, int NTBP, int KIND>
static int launch_hl_mode(const BatchArgsH &a, int blocks, size_t smem, hipStream_t stream) {
    auto go = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * VGHL_WAVES), smem, stream, a);
        return (int)hipGetLastError();
    };
#ifndef VGHL_ONLY_DOT                   // (measurement / disassembly builds: one instantiation per unit)
    if (a.mode == VGH_COS) return go(vg_batch_hl_kernel<VT, NTBP, VGH_COS, KIND>);
    if (a.mode == VGH_L2) return go(vg_batch_hl_kernel<VT, NTBP, VGH_L2, KIND>);
#endif
    return go(vg_batch_hl_kernel<VT, NTBP, VGH_DOT, KIND>);
}
template <int VT, int KIND>
static int launch_hl(const BatchArgsH &a, int ntbp, int blocks, size_t smem, hipStream_t stream) {
#ifdef VGHL_ONLY_NTBP
    return ntbp == VGHL_ONLY_NTBP ? launch_hl_mode<VT, VGHL_ONLY_NTBP, KIND>(a, blocks, smem, stream) : -1;
#else
    if (ntbp == 24) return launch_hl_mode<VT, 24, KIND>(a, blocks, smem, stream);
    if (ntbp == 32) return launch_hl_mode<VT, 32, KIND>(a, blocks, smem, stream);
    return launch_hl_mode<VT, 48, KIND>(a, blocks, smem, stream);
#endif
}
template <int VT, int XU>
static int launch_hlx_mode(const BatchArgsH &a, int sets, int blocks, size_t smem, hipStream_t stream, int subs_given = 0) {
    const int subs = subs_given ? subs_given : VGHL_WAVES / sets;
    if (a.mode == VGH_COS) hipLaunchKernelGGL((vg_batch_hx_kernel<VT, VGH_COS, XU>), dim3((unsigned)blocks), dim3(64 * VGHX_WAVES), smem, stream, a, sets, subs);
    else if (a.mode == VGH_L2) hipLaunchKernelGGL((vg_batch_hx_kernel<VT, VGH_L2, XU>), dim3((unsigned)blocks), dim3(64 * VGHX_WAVES), smem, stream, a, sets, subs);
    else hipLaunchKernelGGL((vg_batch_hx_kernel<VT, VGH_DOT, XU>), dim3((unsigned)blocks), dim3(64 * VGHX_WAVES), smem, stream, a, sets, subs);
    return (int)hipGetLastError();
}
template <int VT>
static int launch_hlx(const BatchArgsH &a, int sets, int blocks, size_t smem, hipStream_t stream, int subs_given = 0) {
    const int xu = (int)((a.xstride / 16 + 63) / 64);                    // 16-byte chunks per lane of the exact evaluation (64 lanes per row)
    if (xu <= 4) return launch_hlx_mode<VT, 4>(a, sets, blocks, smem, stream, subs_given);
    if (xu <= 8) return launch_hlx_mode<VT, 8>(a, sets, blocks, smem, stream, subs_given);
    if (xu <= 12) return launch_hlx_mode<VT, 12>(a, sets, blocks, smem, stream, subs_given);
    return -1;
}

extern "C" int vghl_bound_f16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream);       // -DVGHL_TU=0 (+ the entry point)
extern "C" int vghl_filter_f16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream);      // 1 (+ the exact kernel)
extern "C" int vghl_bound_bf16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream);      // 2
extern "C" int vghl_filter_bf16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream);     // 3
extern "C" int vghl_bound_f32(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream);       // 4
extern "C" int vghl_filter_f32(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream);      // 5
extern "C" int vghl_exact_f16(const BatchArgsH *a, int sets, int blocks, size_t smem, hipStream_t stream);
extern "C" int vghl_exact_bf16(const BatchArgsH *a, int sets, int blocks, size_t smem, hipStream_t stream);
extern "C" int vghl_exact_f32(const BatchArgsH *a, int sets, int blocks, size_t smem, hipStream_t stream);

#if VGHL_TU == 1
extern "C" int vghl_filter_f16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream) { return launch_hl<T_F16, VGH_FILTER>(*a, ntbp, blocks, smem, stream); }
extern "C" int vghl_exact_f16(const BatchArgsH *a, int sets, int blocks, size_t smem, hipStream_t stream) { return launch_hlx<T_F16>(*a, sets, blocks, smem, stream); }
// (vg_batch_q8.hip's long rows: `waves` 32-query regions per query group, one region per block)
extern "C" int vghl_exact_regions_f16(const BatchArgsH *a, int waves, int regions, size_t smem, hipStream_t stream) { return launch_hlx<T_F16>(*a, waves, regions, smem, stream, 1); }
#elif VGHL_TU == 2
extern "C" int vghl_bound_bf16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream) { return launch_hl<T_BF16, VGH_BOUNDK>(*a, ntbp, blocks, smem, stream); }
#elif VGHL_TU == 3
extern "C" int vghl_filter_bf16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream) { return launch_hl<T_BF16, VGH_FILTER>(*a, ntbp, blocks, smem, stream); }
extern "C" int vghl_exact_bf16(const BatchArgsH *a, int sets, int blocks, size_t smem, hipStream_t stream) { return launch_hlx<T_BF16>(*a, sets, blocks, smem, stream); }
extern "C" int vghl_exact_regions_bf16(const BatchArgsH *a, int waves, int regions, size_t smem, hipStream_t stream) { return launch_hlx<T_BF16>(*a, waves, regions, smem, stream, 1); }
#elif VGHL_TU == 4
extern "C" int vghl_bound_f32(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream) { return launch_hl<T_F32, VGH_BOUNDK>(*a, ntbp, blocks, smem, stream); }
#elif VGHL_TU == 5
extern "C" int vghl_filter_f32(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream) { return launch_hl<T_F32, VGH_FILTER>(*a, ntbp, blocks, smem, stream); }
extern "C" int vghl_exact_f32(const BatchArgsH *a, int sets, int blocks, size_t smem, hipStream_t stream) { return launch_hlx<T_F32>(*a, sets, blocks, smem, stream); }
extern "C" int vghl_exact_regions_f32(const BatchArgsH *a, int waves, int regions, size_t smem, hipStream_t stream) { return launch_hlx<T_F32>(*a, waves, regions, smem, stream, 1); }
#else
extern "C" int vghl_bound_f16(const BatchArgsH *a, int ntbp, int blocks, size_t smem, hipStream_t stream) { return launch_hl<T_F16, VGH_BOUNDK>(*a, ntbp, blocks, smem, stream); }

extern "C" int vg_batch_hl_queries_per_block(long long stride_bytes) {
    const int ntbp = vghl_ntbp(stride_bytes);
    return ntbp ? VGHL_QS(ntbp) * VGH_QPW : 0;
}
extern "C" int vg_batch_hl_serves(long long stride_bytes, int k) { return vghl_ntbp(stride_bytes) != 0 && k >= 1 && k <= VGH_MAX_K; }
extern "C" int vg_batch_hl_regions(long long stride_bytes, int nq_pad, int npart) {
    const int qpb = vg_batch_hl_queries_per_block(stride_bytes);
    return qpb ? (nq_pad / qpb) * npart * VGHL_WAVES : 0;
}

extern "C" int vg_batch_merge_launch(const uint64_t *dev_cand, int nq_pad, int lists_per_query, int npart, int k,
                                     uint64_t *dev_out_keys, hipStream_t stream);        // vg_batch.hip

// (float) sum q^2 of every query of a batch (f64 sums, one wavefront per query): what the kernels' gates and the A operand's "multiplies
// as zero" rule read.  -1 for a query the filter cannot judge - Inf / NaN elements, a norm of zero or out of [1e-30, 1e30] - which the
// host answers with a scan of its own; 0 for the padding slots.
template <int TC>
__global__ __launch_bounds__(256) void vg_query_norms_kernel(const uint8_t *queries, long long stride, int dim, int nq_real, int nq_pad, float *out) {
    const int q = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (q >= nq_pad) return;
    double s = 0.0;
    if (q < nq_real) {
        const uint8_t *row = queries + (long long)q * stride;
        for (int e = lane; e < dim; e += 64) {
            float v;
            if constexpr (TC == 2) v = reinterpret_cast<const float *>(row)[e];
            else if constexpr (TC == 0) v = (float)reinterpret_cast<const _Float16 *>(row)[e];
            else v = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(row)[e] << 16);
            s += (double)v * (double)v;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) out[q] = (q >= nq_real) ? 0.0f : ((s >= 1.0e-30 && s <= 1.0e30) ? (float)s : -1.0f);
}
extern "C" int vg_batch_hl_query_norms(const uint8_t *dev_queries, long long stride_bytes, int dim, int type_code, int nq_real, int nq_pad,
                                       float *dev_out, hipStream_t stream) {
    const unsigned blocks = (unsigned)((nq_pad * 64 + 255) / 256);
    if (type_code == 2) hipLaunchKernelGGL(vg_query_norms_kernel<2>, dim3(blocks), dim3(256), 0, stream, dev_queries, stride_bytes, dim, nq_real, nq_pad, dev_out);
    else if (type_code == 1) hipLaunchKernelGGL(vg_query_norms_kernel<1>, dim3(blocks), dim3(256), 0, stream, dev_queries, stride_bytes, dim, nq_real, nq_pad, dev_out);
    else hipLaunchKernelGGL(vg_query_norms_kernel<0>, dim3(blocks), dim3(256), 0, stream, dev_queries, stride_bytes, dim, nq_real, nq_pad, dev_out);
    return (int)hipGetLastError();
}

// dev_rows: the TILE-MAJOR copy of what the matrix core multiplies (type_code 0 / 1: the f16 / bf16 corpus; 2: the bf16 shadow copy of
// an f32 corpus), stride_bytes per row; dev_xrows / xstride_bytes: the row-major corpus the exact evaluation reads; dev_queries: the
// queries in the corpus' own type and stride (zero padded rows, zero rows up to nq_pad); dev_row_nn as for vg_batch_h_launch;
// dev_query_nn: (float) sum q^2 per query, nq_pad of them (0 for the padding).
// dev_pairs / dev_pair_counts: vg_batch_hl_regions() regions of pair_cap pairs + the counts + one overflow word (a region ran full:
// *the caller* reads it back and answers the batch another way).  Returns 0, -1 if the shape is not served, a hipError_t otherwise.
extern "C" int vg_batch_hl_launch(const uint8_t *dev_rows, long long n_rows, long long stride_bytes, int dim, int type_code,
                                  const uint8_t *dev_xrows, long long xstride_bytes,
                                  const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root,
                                  const float *dev_row_nn, const float *dev_query_nn, uint64_t *dev_cand, int npart,
                                  uint64_t *dev_out_keys, unsigned long long *dev_evals,
                                  uint64_t *dev_pairs, uint32_t *dev_pair_counts, int pair_cap, hipStream_t stream) {
    const int ntbp = vghl_ntbp(stride_bytes);
    if (!ntbp || k < 1 || k > VGH_MAX_K || !dev_pairs || !dev_pair_counts || pair_cap < 1) return -1;
    const int sets = VGHL_QS(ntbp), qpb = sets * VGH_QPW;
    if (nq_pad % qpb != 0 || npart < 1 || npart > VG_SEL_MAX_HEADS || n_rows < 1) return -1;
    if (mode < VGH_DOT || mode > VGH_L2 || !dev_row_nn || !dev_query_nn) return -1;
    if (xstride_bytes / 16 > 12 * 64) return -1;
    BatchArgsH a;
    a.rows = dev_rows; a.tiled = 1; a.queries = dev_queries; a.row_nn = dev_row_nn; a.cand = dev_cand;
    a.xrows = dev_xrows; a.xqueries = dev_queries; a.xstride = xstride_bytes;
    a.cerr = (float)(dim + 64) * 4.76837158203125e-7f + (type_code == 2 ? 0.0078125f + 1.52587890625e-5f : 0.0f);
    a.n_rows = n_rows; a.stride = stride_bytes; a.nq_pad = nq_pad; a.nq_real = nq_real; a.npart = npart; a.k = k;
    a.mode = mode; a.root = root; a.dim = dim; a.evals = dev_evals;
    a.pairs = dev_pairs; a.pair_counts = dev_pair_counts; a.pair_cap = pair_cap; a.qnn = dev_query_nn; a.lds_pairs = 0; a.part_group = 0;
    const int G = nq_pad / qpb;
    a.n_regions = G * npart * VGHL_WAVES;
    hipError_t e = hipMemsetAsync(dev_pair_counts + a.n_regions, 0, sizeof(uint32_t), stream);          // the overflow flag
    if (e != hipSuccess) return (int)e;
    const int blocks = G * ((npart + 7) / 8) * 8;
    const size_t smem_exact = (size_t)VGH_QPW * (8 + 4 + 4 + 4) + (size_t)VGH_QPW * k * 8;
    auto launch = [&](const BatchArgsH &b, int kind) -> int {
        const size_t smem = vghl_lds_bytes(ntbp, k, kind == VGH_BOUNDK);
        if (kind == VGH_BOUNDK)
            return type_code == 2 ? vghl_bound_f32(&b, ntbp, blocks, smem, stream)
                                  : (type_code == 1 ? vghl_bound_bf16(&b, ntbp, blocks, smem, stream) : vghl_bound_f16(&b, ntbp, blocks, smem, stream));
        return type_code == 2 ? vghl_filter_f32(&b, ntbp, blocks, smem, stream)
                              : (type_code == 1 ? vghl_filter_bf16(&b, ntbp, blocks, smem, stream) : vghl_filter_f16(&b, ntbp, blocks, smem, stream));
    };
    auto launch_exact = [&](const BatchArgsH &b) -> int {
        const int xblocks = G * b.npart_total * sets;
        return type_code == 2 ? vghl_exact_f32(&b, sets, xblocks, smem_exact, stream)
                              : (type_code == 1 ? vghl_exact_bf16(&b, sets, xblocks, smem_exact, stream) : vghl_exact_f16(&b, sets, xblocks, smem_exact, stream));
    };
    const long long ntiles = (n_rows + VGH_TILE - 1) / VGH_TILE;
    // the bound pre-pass over the first 1 / 512 of the rows, then the filter + exact passes in stages over growing row ranges
    // (vg_batch_common.h) - the split form has no feedback inside a stage: a small corpus gets a small first stage instead
    long long pre = 0;
    {
        const int denom = vg_sw(SW_VG_BATCH_PREPASS, VGB_PREPASS_DENOM_DEFAULT);
        if (denom > 0) pre = ((std::max<long long>(ntiles / denom, 2 * (long long)k) + npart - 1) / npart) * npart;
        if (pre * 2 > ntiles) pre = ntiles;              // a small corpus: bounds over all of it, then ONE filter + exact stage
    }
    int rc;
    a.npart_total = npart; a.part_base = 0; a.seed = 0;
    a.init_keys = nullptr;
    if (pre > 0) {
        a.tile_begin = 0; a.tile_end = pre; a.tiles_per_part = (int)((pre + npart - 1) / npart);
        if ((rc = launch(a, VGH_BOUNDK)) != 0) return rc;
        if ((rc = vg_batch_merge_launch(dev_cand, nq_pad, a.npart_total, npart, k, dev_out_keys, stream)) != 0) return rc;
        a.init_keys = dev_out_keys;
    }
    long long bounds[16];
    const int nstages = vgb_stage_bounds(ntiles, pre, bounds, 16, 200);
    for (int s = 0; s < nstages; ++s) {
        a.tile_begin = bounds[s]; a.tile_end = bounds[s + 1];
        a.tiles_per_part = (int)((a.tile_end - a.tile_begin + npart - 1) / npart);
        a.seed = (s > 0) ? 1 : 0;
        if ((rc = launch(a, VGH_FILTER)) != 0) return rc;
        if ((rc = launch_exact(a)) != 0) return rc;
        if ((rc = vg_batch_merge_launch(dev_cand, nq_pad, a.npart_total, npart, k, dev_out_keys, stream)) != 0) return rc;
        a.init_keys = dev_out_keys;
    }
    return 0;
}
#endif   // VGHL_TU
