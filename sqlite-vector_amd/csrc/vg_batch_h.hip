// vg_batch_h.hip - batched queries over an f16 / bf16 corpus: Q x C^T on the matrix cores
// (v_mfma_f32_32x32x16_f16 / _bf16) as a FILTER, the reference's own f64 arithmetic for what passes it.
//
// The reference accumulates half-precision distances in f64 (distance-avx2.c:166-582) and the single-query kernel
// (vg_scan.h, vg_half.h) follows it; that arithmetic is what bounds those scans (~6 TB/s of 8), and a batch of Q
// queries costs Q of them.  Here the matrix core computes s~ = sum q x for 256 queries x 32 rows at a time: products of
// two halves are exact in f32, the f32 accumulation is off by at most ~D * 2^-24 * |q| |x| (measured behaviour of the
// instruction: tools/mfma_half_probe.hip - subnormal inputs are kept, each instruction rounds one aligned 16-term sum).
// s~ only decides which (query, row) pairs CAN beat the query's current k-th best, with that bound (x4) as slack:
//     dot      -(s~ + E) <= thr                     E = c |q| |x|,  c = (D + 64) * 2^-21
//     cosine   s~ + E >= (1 - thr) |q| |x| (1 - 1e-5)
//     L2       |q|^2 + |x|^2 - 2 (s~ + E) <= thr^2 (1 + 1e-5)         (|x|^2 per row from the corpus' cached vector)
// Every pair that passes is re-evaluated by the whole wavefront with the single-query kernel's accumulator
// (AccumHalf: f32 difference / product widened to f64, f64 sums, the Inf/NaN slow path of vg_half.h), so the distances
// that reach the lists are the single scan's distances; pairs the filter cannot judge (rows or queries with Inf / NaN,
// norms outside [1e-15, 1e15]) always pass.  After the lists have warmed up (two-pass launch as in vg_batch.hip) a
// handful of pairs per query and partition take that path; the rest of the corpus costs one MFMA per 32 x 32 x 16 block.
//
// Skeleton = vg_batch_i8.hip (same operand bytes per lane: 16 bytes = 8 halves of one row per k-step): 8 wavefronts x
// 32 queries stationary in registers, tiles of 32 rows through LDS by LDS-DMA, transposed by 16-byte chunk.  Rows of
// 1 - 2 KiB (up to 1024 elements) run with 4 wavefronts per workgroup: A is then up to 256 registers per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "vg_accum.h"
#include "vg_batch_common.h"
#include "vg_batch_h_defs.h"


#ifndef VGH_WAVES
#define VGH_WAVES 8                     // wavefronts per workgroup for rows up to 1 KiB (two per SIMD) ...
#endif
#define VGH_WAVES_LONG 4                // ... and for rows up to 2 KiB: A alone is up to 256 registers, one wavefront per SIMD
#define VGH_WAVES_OF(NTB) ((NTB) <= 32 ? VGH_WAVES : VGH_WAVES_LONG)
// Rows of up to 768 bytes (NTB <= 24) also come as 4-wavefront workgroups of which TWO share a CU (W = 4): a tile barrier then couples
// four wavefronts instead of eight and the two workgroups' DMA / gate / survivor phases drift apart - 1024 x 10M x 384: f32 dot
// 9.07 -> 8.58 ms, f32 L2 8.67 -> 8.36, f16 dot 8.70 -> 8.61 (profiles/r5t_batch_h_two_4wave_workgroups_per_cu.txt; ONE such workgroup
// per CU: 10.8 ms).  Chosen per launch (vg_batch_h_plan): both workgroups' LDS - tiles + 32 k-key lists per wavefront - must fit.
#define VGH_HAS_W4(NTB) ((NTB) <= 24)
#ifndef VGH_BPIPE
#define VGH_BPIPE 4                     // B-operand ds_read_b128s in flight ahead of the MFMA that consumes them
#endif
#ifndef VGH_ABLATE
#define VGH_ABLATE 0                    // measurement builds (wrong results): 1 = filter computed, survivors dropped; 2 = no filter
#endif

#ifndef VGH_CHAINS
#define VGH_CHAINS 1                    // 2 (with VGH_PIPE): even and odd k steps sum into accumulators of their own.  Measured, same box, 1024 x 10M x 384
#endif                                  // f32 through the bf16 filter: 8.19 ms against 8.05 with one chain (profiles/r6k_*): off
#ifndef VGH_STAGGER
#define VGH_STAGGER 1                   // FILTER kind, 8 wavefronts: the two wavefronts of a SIMD run half a tile apart (see the tile loop)
#endif
#ifndef VGH_PIPE
#define VGH_PIPE 0                      // 1: the gate of tile i-1 runs in the issue gaps of tile i's MFMA chain.  Measured 8.05 ms against 7.95 without
                                        // (same box, profiles/r6k_*): the 16 register copies it needs cost what the gate it hides costs
#endif

#ifndef VGH_TIMING
#define VGH_TIMING 0                    // measurement builds (tools/tools_half_timing.py): where a wavefront's time goes
#endif
#if VGH_TIMING
// (timing builds are single-unit: tools/build_half_variants.sh compiles this file once with all instantiations - see there)
// s_memtime ticks summed over all wavefronts: k loop | filter | survivors | DMA wait | barrier | whole kernel | wave-tiles |
// (exact evaluations << 32) + wave-tiles with survivors
__device__ unsigned long long vgh_ticks[16];  // [8..13] real pass only: pending registers | entries | exact ticks | offer ticks | phase ticks | passing pairs
extern "C" int vg_batch_h_timing(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(vgh_ticks), sizeof(vgh_ticks)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vgh_ticks), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define VGH_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define VGH_TICK(var)
#endif

template <int OFF>
__device__ __forceinline__ void vgh_lds_read128(vgh_i32x4 &dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void vgh_wait_lds(vgh_i32x4 &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}


// NTB = 32-byte k-steps per row (rows up to NTB * 16 elements)
// BOUND = the pre-pass variant: no exact evaluation at all.  A pair that passes the filter enters its list with an UPPER
// BOUND of its distance (the filter's own estimate plus its error bound); the k-th smallest bound of a query is then an
// upper bound of its final k-th best distance - the start threshold of the real pass, which scans every row.
// KIND: VGH_REAL (filter + exact evaluation + lists, all in one kernel), VGH_BOUNDK (the pre-pass above), VGH_FILTER - the filter
// alone: pairs that pass are APPENDED to the wavefront's region of a device buffer and evaluated afterwards by vg_batch_hx_kernel.
// Why: with the exact evaluations (an HBM round trip each, ~3500 cycles, 8 or 4 wavefronts waiting at the tile barrier) and the lists
// gone from the streaming kernel its LDS holds a tile ring NB deep instead of 2 - and the LDS-DMA stream is what bounds this kernel:
// ~25-29 GB/s per CU (profiles/r6b_*: 61 GB per 1024-query batch through LDS in 8.3 ms with two 4-wavefront workgroups per CU, each
// streaming the copy for its own 128 queries).  One 8-wavefront workgroup streams a tile once for 256 queries - half the bytes - and
// with NB - 1 tiles in flight their latency (~2000 cycles each) is covered.
#define VGH_RING_OF(NTB) ((NTB) <= 16 ? 6 : (NTB) <= 24 ? 5 : (NTB) <= 32 ? 4 : (NTB) <= 48 ? 3 : 2)
template <int VT, int NTB, int MODE, int KIND, int W>
__global__ __launch_bounds__(64 * W, (W == 4 && VGH_HAS_W4(NTB) && KIND != VGH_FILTER) ? 2 : 1) void vg_batch_h_kernel(BatchArgsH a) {
    constexpr bool BOUND = (KIND == VGH_BOUNDK), FILT = (KIND == VGH_FILTER);
    constexpr int NB = FILT ? VGH_RING_OF(NTB) : 2;                     // tile buffers in LDS (NB - 1 tiles in flight)
    constexpr int WAVES = W, THREADS = 64 * WAVES, QPB = WAVES * VGH_QPW;
    constexpr int XU = ((NTB <= 32) ? 1 : 2) * (VT == T_F32 ? 2 : 1);    // 16-byte chunks per lane in the exact evaluation
    constexpr bool COS = (MODE == VGH_COS), L2M = (MODE == VGH_L2);
    // VT == T_F32: an f32 corpus.  The matrix core reads a bf16 SHADOW copy of it (BOTH inputs of every product rounded to
    // 8 bits of precision, unit roundoff 2^-8 each: the filter's error bound grows by (2^-7 + 2^-16) |q||x|), the exact
    // evaluation reads the f32 rows with the single-query kernel's f32 arithmetic.
    constexpr bool XF32 = (VT == T_F32);
    constexpr int FT = XF32 ? T_BF16 : VT;                               // element type the matrix core multiplies
    constexpr int ACC = COS ? (XF32 ? A_COS : A_COSN) : (L2M ? A_L2 : A_DOT);   // the exact evaluation's accumulator
    typedef Accum<VT, ACC> Exact;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int TILE_BYTES = NTB * 2 * 512;                           // chunk column c of the 32 rows at c * 512 + row * 16
    uint8_t *tile0 = smem;
    float *rstat_lds = reinterpret_cast<float *>(smem + NB * TILE_BYTES);                // [2 slots][4 tiles][32]: sum x^2
    double *qq_lds = reinterpret_cast<double *>(rstat_lds + 2 * 128);                    // [waves][32]: sum q^2 (f64)
    uint32_t *qsp_lds = reinterpret_cast<uint32_t *>(qq_lds + WAVES * VGH_QPW);       // [waves][32]: query holds Inf / NaN
    float *thr_lds = reinterpret_cast<float *>(qsp_lds + WAVES * VGH_QPW);            // [waves][32]: k-th best so far
    uint64_t *lists = reinterpret_cast<uint64_t *>(thr_lds + WAVES * VGH_QPW);        // [waves][32][k]  (not in the FILTER kind)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31, h = lane >> 5;
    const int k = a.k;

    const int G = a.nq_pad / QPB;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int g = idx % G;
    const int part = (idx / G) * 8 + xcd;
    if (part >= a.npart) return;
    const int q0 = g * QPB + wave * VGH_QPW;
    const int chunks_per_row = (int)(a.stride / 16);
    const int xchunks = (int)(a.xstride / 16);                           // 16-byte chunks of an exact-evaluation row

    // ---- A operand: lane (x, h) keeps bytes [32t + 16h, +16) of query x
    vgh_i32x4 areg[NTB];
    {
        if constexpr (XF32) {                                             // f32 query -> bf16 (round to nearest even) on the fly
            const uint8_t *qrow = a.xqueries + (long long)(q0 + x) * a.xstride;
            auto bf = [](uint32_t lo, uint32_t hi) -> int {
                const uint32_t l = (lo + 0x7FFFu + ((lo >> 16) & 1u)) >> 16, u = (hi + 0x7FFFu + ((hi >> 16) & 1u)) & 0xFFFF0000u;
                return (int)(l | u);
            };
            vgb_static_for<0, NTB>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const int off = (32 * t + 16 * h) * 2;                    // byte offset of the same 8 elements in the f32 row
                uint4 v0 = make_uint4(0u, 0u, 0u, 0u), v1 = make_uint4(0u, 0u, 0u, 0u);
                if (off + 16 <= a.xstride) v0 = *reinterpret_cast<const uint4 *>(qrow + off);
                if (off + 32 <= a.xstride) v1 = *reinterpret_cast<const uint4 *>(qrow + off + 16);
                areg[t] = vgh_i32x4{bf(v0.x, v0.y), bf(v0.z, v0.w), bf(v1.x, v1.y), bf(v1.z, v1.w)};
            });
        } else {
            const uint8_t *qrow = a.queries + (long long)(q0 + x) * a.stride;
            vgb_static_for<0, NTB>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const int off = 32 * t + 16 * h;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (off < a.stride) v = *reinterpret_cast<const uint4 *>(qrow + off);
                areg[t] = vgh_i32x4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            });
        }
    }
    // ---- per-query statistics of the exact path (the single-query kernel's query_stat with 64 lanes per row)
    double *qq_w = qq_lds + wave * VGH_QPW;
    uint32_t *qsp_w = qsp_lds + wave * VGH_QPW;
    float *thr_w = thr_lds + wave * VGH_QPW;
    uint64_t *wave_lists = lists + (size_t)wave * VGH_QPW * k;
    for (int qi = 0; qi < VGH_QPW; ++qi) {
        uint4 qv[XU];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            qv[u] = make_uint4(0u, 0u, 0u, 0u);
            if (lane + 64 * u < xchunks) qv[u] = reinterpret_cast<const uint4 *>(a.xqueries + (long long)(q0 + qi) * a.xstride)[lane + 64 * u];
        }
        if constexpr (XF32) {
            const typename Accum<T_F32, A_COS>::QStat s = Accum<T_F32, A_COS>::template query_stat<XU>(qv, 6);
            if (lane == 0) { qq_w[qi] = (double)s.qq; qsp_w[qi] = 0u; }       // (Inf / NaN queries: the norm check below)
        } else {
            const typename Accum<VT, A_COSN>::QStat s = Accum<VT, A_COSN>::template query_stat<XU>(qv, 6);
            if (lane == 0) { qq_w[qi] = s.qq; qsp_w[qi] = s.qspecial; }
        }
    }
    {   // a query the filter cannot judge (Inf / NaN elements, norm out of range) multiplies as ZERO: its accumulators
        // then hold the "accept everything" start value instead of Inf / NaN, and every row takes the exact path
        const float qqx = (float)qq_w[x];
        if ((qsp_w[x] != 0u) || !(qqx >= VGH_NORM_LO && qqx <= VGH_NORM_HI)) {
#pragma unroll
            for (int t = 0; t < NTB; ++t) areg[t] = vgh_i32x4{0, 0, 0, 0};
        }
    }
    if (lane < VGH_QPW) {                               // thresholds: the pre-pass bound; padding queries never accept
        float t = a.init_keys ? vgb_kth_distance(a.init_keys[(long long)(q0 + lane) * 64 + (k - 1)]) : INFINITY;
        if (q0 + lane >= a.nq_real) t = -INFINITY;
        thr_w[lane] = t;
    }
    if constexpr (!FILT) {
        const bool seeded = a.seed != 0 && part == 0;                  // (exact keys of rows no later stage meets again)
        for (int s = lane; s < VGH_QPW * k; s += 64)
            wave_lists[s] = seeded ? a.init_keys[(long long)(q0 + s / k) * 64 + s % k] : VG_EMPTY_KEY;
    }
    for (int s = tid; s < NB * TILE_BYTES / 4; s += THREADS) reinterpret_cast<uint32_t *>(tile0)[s] = 0u;   // pad columns
    __syncthreads();

    // ---- tile streaming by LDS-DMA (vg_batch_i8.hip): piece p = chunk columns 2p, 2p+1 of all 32 rows
    const int npieces = (chunks_per_row + 1) / 2;
    const long long tile_first = a.tile_begin + (long long)part * a.tiles_per_part;
    const long long tile_last = min(tile_first + a.tiles_per_part, a.tile_end);
    const unsigned long long stride_b = (unsigned long long)a.stride;
    const uint32_t lds_tile0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)tile0;
    // From the tile-major copy the first wavefront of every SIMD (waves 0-3 of an 8-wavefront workgroup: the arbiter favours the
    // older wavefront, they reach the barrier early whatever they do) moves ALL pieces; issuing wavefront w moves the
    // CONTIGUOUS pieces w * NPIECE .. w * NPIECE + NPIECE - 1.  From the tile-major copy a piece is 1 KiB of contiguous memory and up to four
    // of them share one M0 set-up (the instruction offset moves the global and the LDS address alike); from a row-major
    // corpus (the bf16 shadow of an f32 corpus, which the single-query filter scan reads by rows) every lane gathers its
    // 16 bytes of row (l & 31) and a piece is one instruction.
#ifndef VGH_FILTER_ISSUERS
#define VGH_FILTER_ISSUERS 8            // (every wavefront issues its share of a tile's DMA pieces: 8.50 -> 8.33 ms against four issuers)
#endif
    constexpr int NISSUE_T = WAVES == 8 ? (FILT ? VGH_FILTER_ISSUERS : 4) : WAVES;   // issuing wavefronts on the tile-major copy ...
    constexpr int NPIECE = (NTB + NISSUE_T - 1) / NISSUE_T;
    // ... while a row-major gather (measured: 10.5 vs 10.0 ms from four wavefronts) stays spread over all of them
    const int nissue = a.tiled != 0 ? NISSUE_T : WAVES, np_mine = a.tiled != 0 ? NPIECE : (NTB + WAVES - 1) / WAVES;
    // piece p carries data if p < npieces; its second chunk column is a pad column when the chunk count is odd (lanes >= 32 off)
    auto piece_mask_of = [&](int p) -> uint64_t {
        if (wave >= nissue || p >= npieces) return 0ull;
        return (2 * p + 1 < chunks_per_row) ? ~0ull : 0xFFFFFFFFull;
    };
    // (all of this wavefront's pieces whole, from the tile-major copy: the back-to-back path)
    const bool all_full = wave < NISSUE_T && a.tiled != 0 && 2 * (wave * NPIECE + NPIECE) <= chunks_per_row;
    const bool tiled = a.tiled != 0;
    auto lane_offset = [&](long long tile) -> uint32_t {
        if (tiled) return (uint32_t)lane * 16u;                     // (whole tiles exist in the copy; rows past the end are masked later)
        const long long row0 = tile * VGH_TILE;                     // row-major: rows past the end re-read the last row
        const long long last = a.n_rows - 1 - row0;
        const uint32_t xr = (uint32_t)((long long)x < last ? (long long)x : last);
        return xr * (uint32_t)a.stride + (uint32_t)h * 16u;
    };
    // the row norms ride the same pipeline, FOUR tiles (512 bytes) per instruction into a two-slot ring, the issuing
    // wavefronts taking turns
    const uint32_t lds_rstat0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)rstat_lds;
    const uint64_t stat_mask = __ballot(lane < 32);
    const uint32_t stat_goff = (uint32_t)lane * 16u;
    auto dma_stat_group = [&](long long tile4, int slot) {          // tiles tile4 .. tile4 + 3
        const uint8_t *b0 = reinterpret_cast<const uint8_t *>(a.row_nn + tile4 * VGH_TILE);
        const uint32_t d0 = lds_rstat0 + (uint32_t)(slot * 512);
        uint32_t keep;
        uint64_t keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %5\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keep_exec) : "v"(stat_goff), "s"(b0), "s"(d0), "s"(stat_mask) : "memory", "scc");
    };
    auto dma_piece = [&](long long tile, uint32_t lane_goff, int buf, int i) {
        if (i >= np_mine) return;
        const int p = wave * np_mine + i;
        const uint64_t pmask = piece_mask_of(p);
        if (pmask == 0) return;
        const uint8_t *sbase = a.rows + (unsigned long long)(tile * VGH_TILE) * stride_b + (unsigned)p * (tiled ? 1024u : 32u);
        const uint32_t lds_dst = lds_tile0 + (uint32_t)(buf * TILE_BYTES + p * 1024);
        uint32_t keep;
        uint64_t keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_and_b64 exec, exec, %5\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keep_exec) : "v"(lane_goff), "s"(sbase), "s"(lds_dst), "s"(pmask) : "memory", "scc");
    };
    // N (1 .. 4) whole pieces of the tile-major copy starting at piece slot i0, back to back
    auto dma_run = [&](long long tile, uint32_t lane_goff, int buf, auto i0c, auto nc) __attribute__((always_inline)) {
        constexpr int i0 = decltype(i0c)::value, N = decltype(nc)::value;
        const uint8_t *sbase = a.rows + (unsigned long long)(tile * VGH_TILE) * stride_b + (unsigned)(wave * NPIECE + i0) * 1024u;
        const uint32_t lds_dst = lds_tile0 + (uint32_t)(buf * TILE_BYTES + (wave * NPIECE + i0) * 1024);
        uint32_t keep;
        if constexpr (N == 1)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
        else if constexpr (N == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
        else if constexpr (N == 3)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    // run r (pieces 4r .. 4r+3 of this wavefront's share) of a tile
    constexpr int NRUN = (NPIECE + 3) / 4;
    auto dma_share = [&](long long tile, uint32_t lane_goff, int buf, auto rc) __attribute__((always_inline)) {
        constexpr int i0 = 4 * decltype(rc)::value, N = NPIECE - i0 < 4 ? NPIECE - i0 : 4;
        if (all_full) dma_run(tile, lane_goff, buf, std::integral_constant<int, i0>{}, std::integral_constant<int, N>{});
        else vgb_static_for<i0, i0 + N>([&](auto pc) { dma_piece(tile, lane_goff, buf, decltype(pc)::value); });
    };

    // (every lambda of this kernel is always_inline: left to its heuristics the compiler may keep one of the survivor-path
    // lambdas as a real function, which puts everything it captures - gates, A - in scratch memory and makes the tile
    // counter a VGPR, i.e. breaks the "s" operands of the DMA asm)
    // ---- per-register filter state (register r of lane (x, h) belongs to query qi(r, h) = (r&3) + 8*(r>>2) + 4*h).
    // The accumulator of register r STARTS at init_reg[r] and the test after the k loop is
    //     acc[r] + gmul[r] * lane_term >= 0         lane_term: |x| (dot, cosine), (1 - c)/2 |x|^2 (L2)
    //   dot     init = thr (1 + 1e-5) + tiny        gmul = c |q|
    //   cosine  init = 0                            gmul = -((1 - thr) |q| (1 - 1e-5 sgn) - c |q|)
    //   L2      init = (thr2 (1 + 1e-5) - (1 - c) |q|^2) / 2 + tiny    gmul = -1
    // "accept everything" (list not full, query the filter cannot judge) = a huge FINITE init / gmul.
    const float cerr = a.cerr;                                              // halves: (D + 64) * 2^-21; f32 behind bf16: + 2^-7 (1 + 2^-9)
    // Only init_reg / gmul live in registers (A alone takes up to 128 of the 256): the thresholds and the query norms
    // they derive from stay in LDS and are read again when a list changes.
    float init_reg[16], gmul[16];
    const bool l2_root = a.root != 0;
    auto set_gate = [&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float thr = thr_w[qi];
        const float qqf = (float)qq_w[qi];
        const float na = sqrtf(qqf);
        const bool qforce = (qsp_w[qi] != 0u) || !(qqf >= VGH_NORM_LO && qqf <= VGH_NORM_HI);
        const bool open = qforce || !(thr < VGH_ACCEPT);                     // +Inf / NaN threshold: accept everything
        if (COS) {
            const float Gf = (1.0f - thr) * na;
            const float gq = Gf - 1e-5f * fabsf(Gf) - cerr * na;
            init_reg[r] = 1e-30f;
            gmul[r] = open ? VGH_ACCEPT : -gq;
        } else if (L2M) {
            const float thr2 = l2_root ? thr * thr : thr;
            const float v = 0.5f * (thr2 * (1.0f + 1e-5f) - (1.0f - cerr) * qqf) + 1e-30f;
            init_reg[r] = (open || !(v < VGH_ACCEPT)) ? VGH_ACCEPT : v;
            gmul[r] = -1.0f;
        } else {
            init_reg[r] = open ? VGH_ACCEPT : thr + 1e-5f * fabsf(thr) + 1e-30f;
            gmul[r] = qforce ? 0.0f : cerr * na;                            // (na may be NaN / Inf for such a query)
        }
        if (BOUND && open) {                             // the bound pass reads s~ back out of the accumulator: it cannot
            init_reg[r] = COS ? 1e-30f : 0.0f;           // start at 3e38; "accept everything" is a huge multiplier instead
            gmul[r] = VGH_ACCEPT;
        }
        if (thr == -INFINITY) {                          // padding queries never pass
            init_reg[r] = COS ? 0.0f : -VGH_ACCEPT;
            gmul[r] = COS ? -VGH_ACCEPT : (L2M ? -1.0f : 0.0f);
        }
    };
    vgb_static_for<0, 16>([&](auto rc) { set_gate(rc); });
    // The tile boundary's FIRST test uses one multiplier for the lane's 16 registers - the largest: lane_term >= 0, so
    //     max_r (acc[r] + gmul[r] * lane_term)  <=  max_r acc[r] + gmax * lane_term
    // and a tile none of whose pairs can pass is recognised with 16 max operations (8 v_max3) and one fused multiply-add instead of 16
    // of each; the per-register test behind it (rare: a few per cent of the tiles) stays exact.  Queries' multipliers differ by their
    // norms only (dot: c |q|; cosine: threshold and norm; L2: all -1), so the looser test lets few more tiles through.
    float gmax = -VGH_ACCEPT;
    auto refresh_gmax = [&]() __attribute__((always_inline)) {
        gmax = gmul[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) gmax = fmaxf(gmax, gmul[r]);
    };
    refresh_gmax();

    // ---- the exact distance of ONE (query, row) pair, by the whole wavefront (wave-uniform arguments): lane c takes
    // chunk c of the row - the single-query kernel with 64 lanes per row and one chunk per lane
    // (two halves: the loads - unconditional, lanes with nothing to load point at 16 zero bytes - and the arithmetic over what they
    // brought; the deferred form issues the loads one tile before it does the arithmetic)
    auto exact_loads = [&](bool really, int qi_u, uint32_t row_u, uint4 (&qv)[XU], uint4 (&xv)[XU]) __attribute__((always_inline)) {
        const uint8_t *qp = a.xqueries + (long long)(q0 + qi_u) * a.xstride;
        const uint8_t *xp = a.xrows + (unsigned long long)row_u * (unsigned long long)a.xstride;
        const uint8_t *zp = reinterpret_cast<const uint8_t *>(vgh_zero_chunk);
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const bool in = really && (lane + 64 * u < xchunks);
            qv[u] = *reinterpret_cast<const uint4 *>(in ? qp + (long long)(lane + 64 * u) * 16 : zp);
            xv[u] = *reinterpret_cast<const uint4 *>(in ? xp + (long long)(lane + 64 * u) * 16 : zp);
        }
    };
    auto exact_finish = [&](int qi_u, uint32_t row_u, float nn_u, const uint4 (&qv)[XU], const uint4 (&xv)[XU]) __attribute__((always_inline)) -> float {
        const uint8_t *qp = a.xqueries + (long long)(q0 + qi_u) * a.xstride;
        const uint8_t *xp = a.xrows + (unsigned long long)row_u * (unsigned long long)a.xstride;
        typename Exact::QStat qs;
        if constexpr (XF32) qs.qq = (float)qq_w[qi_u];
        else { qs.qq = qq_w[qi_u]; qs.qspecial = qsp_w[qi_u]; }
        Exact acc;
        acc.init();
#pragma unroll
        for (int u = 0; u < XU; ++u) acc.chunk(qv[u], xv[u]);
        float d;
        if constexpr (XF32) {
            d = acc.finish(qs, 6, a.root);                            // the single-query f32 kernel's arithmetic (Inf / NaN propagate)
        } else {
            if constexpr (COS) d = acc.finish_cached_norm(qs, 6, nn_u);
            else d = acc.finish(qs, 6, a.root);
            // Inf / NaN in the row or the query: the reference, replayed (every lane computes the same thing; the flag is
            // the same in all lanes too - as a declared-uniform value it keeps this branch out of the divergence analysis)
            if (__builtin_amdgcn_readfirstlane((int)acc.special(qs, 6)) != 0)
                d = vg_slow_distance<VT, (COS ? A_COS : ACC)>(reinterpret_cast<const uint16_t *>(qp), reinterpret_cast<const uint16_t *>(xp), a.dim, a.root);
        }
        return vg_clamp(d);
    };
    auto exact_distance = [&](int qi_u, uint32_t row_u, float nn_u) __attribute__((always_inline)) -> float {
        uint4 qv[XU], xv[XU];
        exact_loads(true, qi_u, row_u, qv, xv);
        return exact_finish(qi_u, row_u, nn_u, qv, xv);
    };
    bool bound_changed = false;
    unsigned n_exact = 0;                                // exact evaluations of this wavefront (wave-uniform)
#if VGH_TIMING
    unsigned long long tk_cnt = 0, tk_regs = 0, tk_entries = 0, tk_exact = 0, tk_offer = 0, tk_phase = 0, tk_pairs = 0;
#endif
    // an exact distance offered to its query's list; a list that changes refreshes the filter gate of register r_u in the lanes
    // (half hh) that hold the query (r_u is wave-uniform but not a constant here: the deferred pair brings its own)
    auto offer = [&](float de, int qi_u, uint32_t row_u, int hh, int r_u) __attribute__((always_inline)) {
        const float thr_u = thr_w[qi_u];
        // strict: rows arrive in scan order, a row that only ties the k-th best has the larger position and loses
        if (!(de < thr_u)) return;
        uint64_t *list = wave_lists + qi_u * k;
        const float nt = vgb_kth_distance(vgb_list_insert(list, k, lane, vg_make_key(de, row_u)));
        if (nt < thr_u) {                                         // never loosens (a pre-pass bound outlives a not-yet-full list)
            if (lane == 0) thr_w[qi_u] = nt;
            if (h == hh) { vgb_static_for<0, 16>([&](auto rc) { if (r_u == decltype(rc)::value) set_gate(rc); }); refresh_gmax(); }
        }
    };
    // slow path: the pairs of register r that passed the filter
    auto reg_insert = [&](auto rc, float acc_r, long long row, float lane_term, bool force, float nn_row) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int q_lo = (r & 3) + 8 * (r >> 2);
        if constexpr (BOUND) {
            // upper bound of the distance from the filter's estimate s~ (the accumulator minus its start value)
            const int qi_lane = q_lo + 4 * h;
            const float qqf = (float)qq_w[qi_lane], na = sqrtf(qqf), nb = sqrtf(nn_row);
            const float st = acc_r - init_reg[r], E = cerr * na * nb;
            float ub;
            if (COS) ub = fminf(1.0f - st / (na * nb) + cerr + 1e-5f, 2.0f);
            else if (L2M) { const float d2 = fmaxf(qqf + nn_row - 2.0f * st + 2.0f * E + 1e-5f * (qqf + nn_row), 0.0f); ub = l2_root ? sqrtf(d2) : d2; }
            else ub = -st + E;
            ub = ub + 1e-5f * fabsf(ub) + 4.76837158203125e-7f * fabsf(init_reg[r]) + 1e-30f;   // (+ what s~ lost next to the start value)
            const bool qforce = (qsp_w[qi_lane] != 0u) || !(qqf >= VGH_NORM_LO && qqf <= VGH_NORM_HI);
            const bool ok = (row < a.n_rows) && (q0 + qi_lane < a.nq_real) && !force && !qforce && (ub < thr_w[qi_lane]);
            if (__ballot(ok) != 0) {
                // only the SMALLEST bound of each query in this tile enters its list: k entries then stand for k
                // different rows all the same, the k-th smallest of them is still an upper bound of the final k-th best,
                // and a list costs one insert per (query, tile) instead of one per passing row (half the inserts of
                // the warm-up; each is an LDS round trip of the whole wavefront)
                uint64_t key = ok ? vg_make_key(ub, (uint32_t)row) : VG_EMPTY_KEY;
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
                    const float nt = vgb_kth_distance(vgb_list_insert(wave_lists + qi_u * k, k, lane, c));
                    if (lane == 0) thr_w[qi_u] = nt;                 // (a list's k-th bound never grows)
                }
                bound_changed = true;                                // the filter's gates are refreshed once per tile
            }
            return;
        }
    };

    if (tile_first < tile_last) {
        vgb_static_for<0, NB - 1>([&](auto jc) {                     // tiles first .. first + NB - 2 into buffers 0 .. NB - 2
            constexpr int j = decltype(jc)::value;
            const long long tj = min(tile_first + j, tile_last - 1);
            const uint32_t goffj = lane_offset(tj);
            vgb_static_for<0, NRUN>([&](auto rc) { dma_share(tj, goffj, j, rc); });
        });
        if (wave == 0) dma_stat_group(tile_first, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // an issuing wavefront that moves exactly NPIECE whole pieces per tile (the tile-major copy, its share full) may leave the pieces
    // of the NB - 2 youngest tiles in flight at a tile's end: loads return in order, so "at most (NB - 2) * NPIECE outstanding" means
    // the next tile's have landed.  Every other wavefront waits for everything it has issued.
    const bool counted_wait = NB > 2 && all_full;
    // FILTER kind: this wavefront's region of the pair buffer
    const long long region = FILT ? ((long long)(g * a.npart_total + a.part_base + part) * WAVES + wave) : 0;
    uint64_t *my_pairs = FILT ? a.pairs + region * a.pair_cap : nullptr;
    unsigned n_pairs = 0;                                            // (wave-uniform)

    constexpr int BP = VGH_BPIPE < NTB ? VGH_BPIPE : NTB;
    vgh_i32x4 bq[BP];
#if VGH_TIMING
    unsigned long long tk_loop = 0, tk_gate = 0, tk_surv = 0, tk_dma = 0, tk_bar = 0;
    const unsigned long long tk_begin = __builtin_readcyclecounter();
#endif
    // PIPE (real passes): the gate of tile i-1 - one fused multiply-add + max per accumulator register - is spread over the issue
    // gaps of tile i's MFMA chain (its own accumulators are long complete: no wait for the chain to drain, no VALU block between two
    // k loops during which the SIMD's matrix pipe idles), its survivors follow that k loop.  One more trip of the loop gates the last
    // tile (its k loop runs over a buffer that is not used).  The bound passes read s~ back out of the accumulator against the
    // CURRENT start value and keep the tile-by-tile form.
    constexpr bool PIPE = (VGH_PIPE != 0) && !BOUND;
    constexpr bool STAGGER = (VGH_STAGGER != 0) && FILT && WAVES == 8;
    constexpr int GPS = (16 + NTB - 1) / NTB;                         // gate registers per k step
    vgh_f32x16 accp;                                                  // PIPE: the accumulators of the tile in front
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[r] = 0.0f;
    float nn_p = 0.0f;
    long long row_p = 0;
    bool have_p = false;
    const long long tile_stop = (PIPE && tile_first < tile_last) ? tile_last + 1 : tile_last;
    for (long long tile = tile_first; tile < tile_stop; ++tile) {
        VGH_TICK(t0);
        const long long ti = tile - tile_first;
        const int cur_buf = (int)(ti % NB), fill_buf = (int)((ti + NB - 1) % NB);
        const long long tile_next = min(tile + NB - 1, tile_last - 1);        // the tile whose DMA this trip issues
        const uint32_t goff_next = lane_offset(tile_next);
        const bool stat_turn = ((ti + 1) & 3) == 0 && wave == (int)(((ti + 1) >> 2) & (NISSUE_T - 1));
        const long long row_cur = tile * VGH_TILE + x;
        // the tile under the gate: this one, or (PIPE) the one in front
        const bool force_p = !(nn_p <= VGH_NORM_HI) || (nn_p < VGH_NORM_LO && nn_p != 0.0f);
        const float lane_term_p = force_p ? 0.0f : (L2M ? 0.5f * (1.0f - cerr) * nn_p : sqrtf(nn_p));
        float margin = -INFINITY;

        // Two accumulator chains (CHAINS = 2): even k steps into acc (which starts at the gate's start values), odd ones into acc1 (zero);
        // their sum is formed where PIPE copies the accumulators to accp anyway.  Between two MFMAs on the SAME accumulator every other
        // instruction - the B-operand read and its wait, a DMA piece, the pipelined gate - breaks the back-to-back issue
        // (MI355X_MICROARCH.md: +43 cycles for the first extra issue slot); with two chains the next MFMA never waits for the last one.
        constexpr bool TWO = (VGH_CHAINS == 2) && PIPE && NTB >= 2;
        vgh_f32x16 acc, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = init_reg[r]; acc1[r] = 0.0f; }
        const uint32_t baddr = lds_tile0 + (uint32_t)(cur_buf * TILE_BYTES + h * 512 + x * 16);
        vgb_static_for<0, BP>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            vgh_lds_read128<1024 * t>(bq[t], baddr);
        });
        vgb_static_for<0, NTB>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int in_flight_after = (NTB - 1 - t) < (BP - 1) ? (NTB - 1 - t) : (BP - 1);
            vgh_wait_lds<in_flight_after>(bq[t % BP]);
            const vgh_i32x4 b = bq[t % BP];
            if constexpr (TWO && (t & 1)) acc1 = vgh_mfma<FT>(areg[t], b, acc1);
            else acc = vgh_mfma<FT>(areg[t], b, acc);
            if constexpr (t + BP < NTB && VGH_ABLATE < 5) vgh_lds_read128<1024 * (t + BP)>(bq[t % BP], baddr);
            // the next tile's DMA (runs of up to four pieces) over the first half of the k loop; every fourth tile one of
            // the issuing wavefronts adds the row norms of the four tiles after this one
            constexpr int NTD = (NTB + 1) / 2;
            vgb_static_for<0, NRUN>([&](auto rc) {
                if constexpr (decltype(rc)::value * NTD / NRUN == t && VGH_ABLATE < 3) dma_share(tile_next, goff_next, fill_buf, rc);
            });
            if constexpr (t == NTD) { if (stat_turn) dma_stat_group(tile + 1, (int)(((ti + 1) >> 2) & 1)); }
            if constexpr (PIPE) {                                    // the gate of the tile in front, GPS registers per k step
                vgb_static_for<t * GPS, ((t + 1) * GPS < 16 ? (t + 1) * GPS : 16)>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    margin = fmaxf(margin, accp[r]);
                });
            }
        });
        float nn_row = rstat_lds[((ti >> 2) & 1) * 128 + (ti & 3) * 32 + x];      // landed with its group of four tiles
        // STAGGER: wavefronts 4-7 (the SECOND wavefront of every SIMD) meet the tile's barrier HERE, right behind their k loop, the
        // first four at the end of the trip.  Past the first tile the halves therefore run half a trip apart: while one wavefront of a
        // SIMD is in its k loop the other one gates, appends, waits for the DMA - the matrix pipe no longer idles while all eight do
        // that together (lock step: 1240 of a tile's 3060 cycles, profiles/r6c_*).  The buffer rules hold: every wavefront has read
        // tile i when it arrives at barrier i, and the issuing wavefronts (0-3) wait for tile i+1's pieces in front of theirs.
        // (from a row-major corpus the second half issues DMA pieces too: they must have landed before ITS barrier)
        if constexpr (STAGGER) {
            if (wave >= WAVES / 2) {
                if (counted_wait && !stat_turn) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NB - 2) * NPIECE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (VGH_ABLATE < 4) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        if constexpr (XF32) nn_row = nn_row * nn_row;                // (the f32 corpus caches ||x||, not sum x^2)
#if VGH_TIMING
        if (!PIPE) asm volatile("s_nop 0" :: "v"(acc[15]));           // the k loop's last MFMA has retired
#endif
        VGH_TICK(t1);

        // ---- tile boundary: one fused multiply-add + max per register, one ballot
        // (a row of ZEROS is judged like any other: every product is exactly 0 and so is the bound's |x| term - a corpus with empty vectors
        // would otherwise pay 256 exact evaluations per such row and workgroup; cosine gives it the reference's distance 1.0 below)
        const bool force = PIPE ? force_p : (!(nn_row <= VGH_NORM_HI) || (nn_row < VGH_NORM_LO && nn_row != 0.0f));   // NaN / Inf / out of range
        const float lane_term = PIPE ? lane_term_p : (force ? 0.0f : (L2M ? 0.5f * (1.0f - cerr) * nn_row : sqrtf(nn_row)));
        if constexpr (!PIPE) {
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                margin = fmaxf(margin, acc[r]);
            });
        }
        if (VGH_ABLATE >= 2) asm volatile("" :: "v"(acc[0]), "v"(acc[15]), "v"(acc1[0]), "v"(acc1[15]));   // (measurement builds without the gate: keep the MFMA chain alive)
        unsigned pend = 0;                                           // bound passes: registers with a passing lane; real passes: "any"
        uint32_t mybits = 0u;                                        // real passes: bit r = this lane's pair of register r passed the filter
        const bool gate_live = PIPE ? have_p : true;
        const long long row_g = PIPE ? row_p : row_cur;              // the row this lane's pairs under the gate belong to
        const float nn_g = PIPE ? nn_p : nn_row;
        if (VGH_ABLATE < 2 && gate_live && __ballot(force || fmaf(gmax, lane_term, margin) >= 0.0f) != 0) {
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                bool pass_r = force || fmaf(gmul[r], lane_term, (PIPE ? accp[r] : acc[r])) >= 0.0f;
                if constexpr (COS) {                                 // a zero-norm row: distance 1.0 whatever the query (distance-cpu.c:74-110)
                    if ((PIPE ? nn_p : nn_row) == 0.0f) pass_r = thr_w[(r & 3) + 8 * (r >> 2) + 4 * h] >= 0.99999f;
                }
                if constexpr (BOUND) pend |= __ballot(pass_r) ? (1u << r) : 0u;
                else mybits |= pass_r ? (1u << r) : 0u;
            });
            if constexpr (!BOUND) {
                if (!(row_g < a.n_rows)) mybits = 0u;
                pend = __ballot(mybits != 0u) != 0ull ? 1u : 0u;
            }
        }
        if (VGH_ABLATE == 1) { if (pend) asm volatile("" :: "s"(pend)); pend = 0; mybits = 0u; }
        VGH_TICK(t2);
#if VGH_TIMING
        if (pend) tk_cnt += 1;
#endif
        if constexpr (BOUND) {
            if (pend) {
                vgb_static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if (pend & (1u << r)) reg_insert(rc, acc[r], row_cur, lane_term, force, nn_row);
                });
            }
            if (bound_changed) { vgb_static_for<0, 16>([&](auto rc) { set_gate(rc); }); refresh_gmax(); bound_changed = false; }
        } else {
            // Real passes: ONE loop over the passing pairs, lanes ascending (a query's rows ascending: scan order) - the exact
            // evaluation and the list insert exist once in the code instead of once per accumulator register.
            if (pend) {
                unsigned long long any;
                while ((any = __ballot(mybits != 0u)) != 0ull) {
                    const int src = __ffsll((long long)any) - 1;
                    const uint32_t bits_u = (uint32_t)__builtin_amdgcn_readlane((int)mybits, src);
                    const int r_u = __ffs((int)bits_u) - 1;
                    if (lane == src) mybits &= mybits - 1u;
                    const int hh = src >> 5, qi_u = (r_u & 3) + 8 * (r_u >> 2) + 4 * hh;
                    if (q0 + qi_u >= a.nq_real) continue;
                    const uint32_t row_u = (uint32_t)__builtin_amdgcn_readlane((int)row_g, src);
                    const float nn_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nn_g), src));
                    ++n_exact;
                    if constexpr (FILT) {
                        if (n_pairs < (unsigned)a.pair_cap) { if (lane == 0) my_pairs[n_pairs] = ((uint64_t)(uint32_t)qi_u << 32) | row_u; }
                        else if (lane == 0) a.pair_counts[a.n_regions] = 1u;     // region full: the host repeats the batch (fused kernel)
                        ++n_pairs;
                    } else {
                        VGH_TICK(te0);
                        // every lane holds the same value (butterfly sums): say so, or the branch in offer() counts as divergent
                        const float de = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, exact_distance(qi_u, row_u, nn_u))));
#if VGH_TIMING
                        tk_cnt += 1ull << 32;
                        const unsigned long long te1 = __builtin_readcyclecounter();
                        tk_exact += te1 - te0; tk_pairs += 1;
#endif
                        offer(de, qi_u, row_u, hh, r_u);
                    }
                }
            }
        }
        VGH_TICK(t3);
#if VGH_TIMING
        if (!BOUND && pend) { tk_entries += 1; tk_phase += t3 - t2; }
#endif
        if (counted_wait && !stat_turn) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NB - 2) * NPIECE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the row norms it issued are its youngest load)
        if constexpr (PIPE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accp[r] = TWO ? acc[r] + acc1[r] : acc[r];
            nn_p = nn_row; row_p = row_cur; have_p = tile < tile_last;
        }
        VGH_TICK(t4);
        if constexpr (STAGGER) { if (wave < WAVES / 2) { asm volatile("" ::: "memory"); if (VGH_ABLATE < 4) __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } }
        else __syncthreads();
#if VGH_TIMING
        const unsigned long long t5 = __builtin_readcyclecounter();
        tk_loop += t1 - t0; tk_gate += t2 - t1; tk_surv += t3 - t2; tk_dma += t4 - t3; tk_bar += t5 - t4;
#endif
    }
#if VGH_TIMING
    if (lane == 0) {
        atomicAdd(&vgh_ticks[0], tk_loop); atomicAdd(&vgh_ticks[1], tk_gate); atomicAdd(&vgh_ticks[2], tk_surv);
        atomicAdd(&vgh_ticks[3], tk_dma); atomicAdd(&vgh_ticks[4], tk_bar);
        atomicAdd(&vgh_ticks[5], __builtin_readcyclecounter() - tk_begin);
        atomicAdd(&vgh_ticks[6], (unsigned long long)(tile_last - tile_first));
        atomicAdd(&vgh_ticks[7], tk_cnt);
        atomicAdd(&vgh_ticks[8], tk_regs); atomicAdd(&vgh_ticks[9], tk_entries); atomicAdd(&vgh_ticks[10], tk_exact);
        atomicAdd(&vgh_ticks[12], tk_phase); atomicAdd(&vgh_ticks[13], tk_pairs);
    }
#endif

    if constexpr (FILT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (no LDS-DMA of the ring may land after this workgroup's LDS is gone)
        if (lane == 0) a.pair_counts[region] = n_pairs < (unsigned)a.pair_cap ? n_pairs : (unsigned)a.pair_cap;
        return;
    }
    for (int s = lane; s < VGH_QPW * 64; s += 64) {
        const int qi = s >> 6, slot = s & 63;
        a.cand[((long long)(q0 + qi) * a.npart_total + a.part_base + part) * 64 + slot] = (slot < k) ? wave_lists[qi * k + slot] : VG_EMPTY_KEY;
    }
    if (!BOUND && a.evals && lane == 0 && n_exact) atomicAdd(a.evals, (unsigned long long)n_exact);
}

// ---- host side
// The kernel instantiations are split over three translation units compiled from this file (build.py): the real-pass
// kernels for f16 here (with the host entry points), for bf16 with -DVGH_TU=1, for f32 corpora with -DVGH_TU=3, the
// bound-pass kernels with -DVGH_TU=2 (f16), 4 (bf16), 5 (f32 corpora) - one unit for all three was the build's critical path
// (-DVGH_TU_ALL: everything in this one unit - the measurement builds of tools/build_half_variants.sh).
#ifndef VGH_TU
#define VGH_TU 0
#endif
extern "C" int vgh_launch_real_f16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_real_bf16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_real_f32(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);   // -DVGH_TU=3
extern "C" int vgh_launch_bound_f16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);    // -DVGH_TU=2
extern "C" int vgh_launch_bound_bf16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);   // -DVGH_TU=4
extern "C" int vgh_launch_bound_f32(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);    // -DVGH_TU=5

template <int VT, int NTB, int MODE, int BOUND, int W>
static int launch_h(const BatchArgsH &a, int blocks, size_t smem, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vg_batch_h_kernel<VT, NTB, MODE, BOUND, W>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((vg_batch_h_kernel<VT, NTB, MODE, BOUND, W>), dim3((unsigned)blocks), dim3(64 * W), smem, stream, a);
    return (int)hipGetLastError();
}
template <int VT, int NTB, int BOUND, int W>
static int launch_h_mode(const BatchArgsH &a, int blocks, size_t smem, hipStream_t stream) {
    if (a.mode == VGH_COS) return launch_h<VT, NTB, VGH_COS, BOUND, W>(a, blocks, smem, stream);
    if (a.mode == VGH_L2) return launch_h<VT, NTB, VGH_L2, BOUND, W>(a, blocks, smem, stream);
    return launch_h<VT, NTB, VGH_DOT, BOUND, W>(a, blocks, smem, stream);
}
template <int VT, int NTB, int BOUND>
static int launch_h_waves(const BatchArgsH &a, int waves, int blocks, size_t smem, hipStream_t stream) {
    if constexpr (VGH_HAS_W4(NTB) && BOUND != VGH_FILTER) {
        if (waves == 4) return launch_h_mode<VT, NTB, BOUND, 4>(a, blocks, smem, stream);
    }
    if (waves != VGH_WAVES_OF(NTB)) return -1;
    return launch_h_mode<VT, NTB, BOUND, VGH_WAVES_OF(NTB)>(a, blocks, smem, stream);
}
template <int VT, int BOUND>
static int launch_h_ntb(const BatchArgsH &a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
#ifdef VGH_ONLY_NTB                     // measurement builds: one row length only (a unit compiles in a minute instead of eight)
    if (ntb != VGH_ONLY_NTB) return -1;
    return launch_h_waves<VT, VGH_ONLY_NTB, BOUND>(a, waves, blocks, smem, stream);
#else
    if (ntb == 8) return launch_h_waves<VT, 8, BOUND>(a, waves, blocks, smem, stream);
    if (ntb == 16) return launch_h_waves<VT, 16, BOUND>(a, waves, blocks, smem, stream);
    if (ntb == 24) return launch_h_waves<VT, 24, BOUND>(a, waves, blocks, smem, stream);
    if (ntb == 32) return launch_h_waves<VT, 32, BOUND>(a, waves, blocks, smem, stream);
    if (ntb == 48) return launch_h_waves<VT, 48, BOUND>(a, waves, blocks, smem, stream);
    return launch_h_waves<VT, 64, BOUND>(a, waves, blocks, smem, stream);
#endif
}

#if VGH_TU == 1 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_real_bf16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_BF16, VGH_REAL>(*a, ntb, waves, blocks, smem, stream);
}
#endif
#if VGH_TU == 2 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_bound_f16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_F16, VGH_BOUNDK>(*a, ntb, waves, blocks, smem, stream);
}
#endif
#if VGH_TU == 4 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_bound_bf16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_BF16, VGH_BOUNDK>(*a, ntb, waves, blocks, smem, stream);
}
#endif
#if VGH_TU == 5 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_bound_f32(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_F32, VGH_BOUNDK>(*a, ntb, waves, blocks, smem, stream);
}
#endif
#if VGH_TU == 3 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_real_f32(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_F32, VGH_REAL>(*a, ntb, waves, blocks, smem, stream);
}
#endif
// the split form: the FILTER kind of the streaming kernel + the exact-evaluation kernel, one unit per element type (-DVGH_TU=6 / 7 / 8)
template <int VT, int XU>
static int launch_hx_mode(const BatchArgsH &a, int waves, int regions, size_t smem, hipStream_t stream) {
    if (a.mode == VGH_COS) hipLaunchKernelGGL((vg_batch_hx_kernel<VT, VGH_COS, XU>), dim3((unsigned)regions), dim3(64 * VGHX_WAVES), smem, stream, a, waves, 1);
    else if (a.mode == VGH_L2) hipLaunchKernelGGL((vg_batch_hx_kernel<VT, VGH_L2, XU>), dim3((unsigned)regions), dim3(64 * VGHX_WAVES), smem, stream, a, waves, 1);
    else hipLaunchKernelGGL((vg_batch_hx_kernel<VT, VGH_DOT, XU>), dim3((unsigned)regions), dim3(64 * VGHX_WAVES), smem, stream, a, waves, 1);
    return (int)hipGetLastError();
}
template <int VT>
static int launch_hx(const BatchArgsH &a, int ntb, int waves, int regions, size_t smem, hipStream_t stream) {
    const int xu = ((ntb <= 32) ? 1 : 2) * (VT == T_F32 ? 2 : 1);       // (the fused kernel's XU)
    if (xu == 1) return launch_hx_mode<VT, 1>(a, waves, regions, smem, stream);
    if (xu == 2) return launch_hx_mode<VT, 2>(a, waves, regions, smem, stream);
    return launch_hx_mode<VT, 4>(a, waves, regions, smem, stream);
}
#if VGH_TU == 6 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_filter_f16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_F16, VGH_FILTER>(*a, ntb, waves, blocks, smem, stream);
}
extern "C" int vgh_launch_exact_f16(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream) {
    return launch_hx<T_F16>(*a, ntb, waves, regions, smem, stream);
}
#endif
#if VGH_TU == 7 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_filter_bf16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_BF16, VGH_FILTER>(*a, ntb, waves, blocks, smem, stream);
}
extern "C" int vgh_launch_exact_bf16(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream) {
    return launch_hx<T_BF16>(*a, ntb, waves, regions, smem, stream);
}
#endif
#if VGH_TU == 8 || defined(VGH_TU_ALL)
extern "C" int vgh_launch_filter_f32(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_F32, VGH_FILTER>(*a, ntb, waves, blocks, smem, stream);
}
extern "C" int vgh_launch_exact_f32(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream) {
    return launch_hx<T_F32>(*a, ntb, waves, regions, smem, stream);
}
#endif
#if VGH_TU == 0
extern "C" int vgh_launch_filter_f16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_filter_bf16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_filter_f32(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_exact_f16(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_exact_bf16(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_exact_f32(const BatchArgsH *a, int ntb, int waves, int regions, size_t smem, hipStream_t stream);
extern "C" int vgh_launch_real_f16(const BatchArgsH *a, int ntb, int waves, int blocks, size_t smem, hipStream_t stream) {
    return launch_h_ntb<T_F16, VGH_REAL>(*a, ntb, waves, blocks, smem, stream);
}

extern "C" int vg_batch_merge_launch(const uint64_t *dev_cand, int nq_pad, int lists_per_query, int npart, int k,
                                     uint64_t *dev_out_keys, hipStream_t stream);        // vg_batch.hip


static int vgh_ntb(long long stride_bytes) {
    const int ntb = (int)((stride_bytes + 31) / 32);
    if (ntb <= 8) return 8;
    if (ntb <= 16) return 16;
    if (ntb <= 24) return 24;
    if (ntb <= 32) return 32;
    if (ntb <= 48) return 48;                                     // rows up to 768 / 1024 elements: 4-wavefront workgroups
    if (ntb <= 64) return 64;
    return 0;
}

static size_t vgh_lds_bytes(int NTB, int k, int waves) {
    return (size_t)2 * NTB * 1024 + 1024 + (size_t)waves * VGH_QPW * (8 + 4 + 4) + (size_t)waves * VGH_QPW * k * 8;
}
// What a batch of nq queries over rows of stride_bytes is launched with: wavefronts per workgroup (= 32 queries each) and the
// workgroups per CU the partition count should aim at.  Two 4-wavefront workgroups per CU where that form exists (NTB <= 24), both
// fit the CU's LDS (k <= 27 at 768-byte rows) and the batch fills at least two of them per partition (nq > 128); VG_BATCH_H_WAVES=8
// keeps round 3's single 8-wavefront workgroup, =4 forces the new form wherever it is instantiated and fits.
// The split form (filter kernel + exact-evaluation kernel, see the FILTER kind) serves every shape the fused kernel serves:
// VG_BATCH_H_SPLIT=1 switches it on.
extern "C" int vg_batch_h_split(long long stride_bytes, int k) {
    const int NTB = vgh_ntb(stride_bytes);
    if (!NTB || k < 1 || k > VGH_MAX_K) return 0;
    return vg_sw(SW_VG_BATCH_H_SPLIT, 0) != 0;                         // opt-in: the two forms measure equal (7.98 against 7.95 ms, profiles/r6k_*)
}
extern "C" int vg_batch_h_plan(long long stride_bytes, int k, int nq, int *waves, int *blocks_per_cu) {
    const int NTB = vgh_ntb(stride_bytes);
    if (!NTB || k < 1 || k > VGH_MAX_K) return -1;
    int w = VGH_WAVES_OF(NTB), bpc = 1;
    if (vg_batch_h_split(stride_bytes, k)) {                     // one workgroup per CU: a tile is streamed once for all its queries
        if (waves) *waves = w;
        if (blocks_per_cu) *blocks_per_cu = 1;
        return 0;
    }
    const int forced = vg_sw(SW_VG_BATCH_H_WAVES, 0);
    if (VGH_HAS_W4(NTB) && forced != 8 && 2 * vgh_lds_bytes(NTB, k, 4) + 4096 <= (size_t)160 * 1024 && (nq > 4 * VGH_QPW || forced == 4)) { w = 4; bpc = 2; }
    if (vgh_lds_bytes(NTB, k, w) > (size_t)160 * 1024) return -1;                 // (2 KiB rows with k >= 30: not served, as vg_batch_h_lds_bytes says)
    if (waves) *waves = w;
    if (blocks_per_cu) *blocks_per_cu = bpc;
    return 0;
}
extern "C" int vg_batch_h_queries_per_block(long long stride_bytes) { return VGH_WAVES_OF(vgh_ntb(stride_bytes)) * VGH_QPW; }

extern "C" size_t vg_batch_h_lds_bytes(long long stride_bytes, int k) {          // (the default 8-wavefront form: the larger of the two)
    const int NTB = vgh_ntb(stride_bytes);
    if (!NTB || k < 1 || k > VGH_MAX_K) return 0;
    const size_t b = vgh_lds_bytes(NTB, k, VGH_WAVES_OF(NTB));
    return b <= 160 * 1024 ? b : 0;
}

// rows [row0, row0 + n) of a row-major corpus -> the TILE-MAJOR copy the batch kernels stream (one 16-byte chunk per thread):
// chunk c of row R goes to tile (R / 32) * (32 * stride) + c * 512 + (R % 32) * 16
__global__ __launch_bounds__(256) void vg_tile_major_kernel(const uint8_t *rows, long long row0, long long n, long long stride, uint8_t *out) {
    const int nch = (int)(stride / 16);
    const long long total = n * nch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long R = row0 + i / nch;
        const int c = (int)(i % nch);
        *reinterpret_cast<uint4 *>(out + (R >> 5) * (32 * stride) + (long long)c * 512 + (R & 31) * 16) =
            *reinterpret_cast<const uint4 *>(rows + R * stride + 16 * c);
    }
}
extern "C" int vg_tile_major_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, uint8_t *dev_out, hipStream_t stream) {
    if (n <= 0) return 0;
    long long blocks = (n * (stride / 16) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(vg_tile_major_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride, dev_out);
    return (int)hipGetLastError();
}

// f32 rows -> their bf16 shadow copy (round to nearest even; zero padded to the shadow stride): 8 elements per thread
__global__ __launch_bounds__(256) void vg_f32_to_bf16_kernel(const uint8_t *rows, long long row0, long long n, long long stride,
                                                             int dim, uint8_t *out, long long ostride) {
    const int groups = (int)(ostride / 16);                               // 16-byte groups (8 elements) per shadow row
    const long long total = n * groups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = row0 + i / groups;
        const int g = (int)(i % groups);
        const float *src = reinterpret_cast<const float *>(rows + r * stride) + 8 * g;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 8 * g + 2 * j;
            const uint32_t lo = (e < dim) ? __float_as_uint(src[2 * j]) : 0u, hi = (e + 1 < dim) ? __float_as_uint(src[2 * j + 1]) : 0u;
            w[j] = ((lo + 0x7FFFu + ((lo >> 16) & 1u)) >> 16) | ((hi + 0x7FFFu + ((hi >> 16) & 1u)) & 0xFFFF0000u);
        }
        *reinterpret_cast<uint4 *>(out + r * ostride + 16 * g) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
extern "C" int vg_f32_to_bf16_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int dim,
                                     uint8_t *dev_out, long long ostride, hipStream_t stream) {
    if (n <= 0) return 0;
    long long blocks = (n * (ostride / 16) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(vg_f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride, dim, dev_out, ostride);
    return (int)hipGetLastError();
}

// f32 rows -> the TILE-MAJOR bf16 shadow copy in one pass (what vg_tile_major_kernel makes of an f16 / bf16 corpus): tile t = rows
// 32t .. 32t+31, chunk column g of the 32 rows = one 512-byte run.  The batched filter streams this copy with 1 KiB-contiguous
// LDS-DMA pieces instead of gathering 32-byte runs from 32 shadow rows (round 2: 1.3x the shadow copy's bytes in HBM traffic).
__global__ __launch_bounds__(256) void vg_f32_to_bf16_tm_kernel(const uint8_t *rows, long long row0, long long n, long long stride,
                                                                int dim, uint8_t *out, long long ostride) {
    const int groups = (int)(ostride / 16);
    const long long tiles = ((row0 + n + 31) >> 5) - (row0 >> 5);         // tiles the rows [row0, row0 + n) touch
    const long long total = tiles * 32 * groups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // consecutive threads take consecutive ROWS of one group column: their 16-byte stores are one contiguous 512-byte run
        const long long tile_rel = i / (32ll * groups);
        const int within = (int)(i - tile_rel * 32ll * groups);
        const int g = within >> 5, rr = within & 31;
        const long long R = (row0 & ~31ll) + tile_rel * 32 + rr;
        if (R < row0 || R >= row0 + n) continue;
        const float *src = reinterpret_cast<const float *>(rows + R * stride) + 8 * g;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 8 * g + 2 * j;
            const uint32_t lo = (e < dim) ? __float_as_uint(src[2 * j]) : 0u, hi = (e + 1 < dim) ? __float_as_uint(src[2 * j + 1]) : 0u;
            w[j] = ((lo + 0x7FFFu + ((lo >> 16) & 1u)) >> 16) | ((hi + 0x7FFFu + ((hi >> 16) & 1u)) & 0xFFFF0000u);
        }
        *reinterpret_cast<uint4 *>(out + (R >> 5) * (32 * ostride) + (long long)g * 512 + (R & 31) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
extern "C" int vg_f32_to_bf16_tm_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int dim,
                                        uint8_t *dev_out, long long ostride, hipStream_t stream) {
    if (n <= 0) return 0;
    const long long tiles = ((row0 + n + 31) >> 5) - (row0 >> 5);
    long long blocks = (tiles * 32 * (ostride / 16) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(vg_f32_to_bf16_tm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride, dim, dev_out, ostride);
    return (int)hipGetLastError();
}

// type_code 0 / 1: dev_rows / dev_queries hold f16 / bf16 elements, zero padded rows of stride_bytes; dev_xrows = the row-major
// corpus; rows_tiled: dev_rows is its tile-major copy (vg_tile_major_launch; whole tiles allocated).
// type_code 2: an f32 corpus - dev_rows is its bf16 shadow copy (stride_bytes per row), dev_xrows / dev_queries the f32 rows /
// queries (xstride_bytes per row).  dev_row_nn: (float) sum x^2 per row (f32: ||x||), readable for 32 floats past the last
// whole tile.  Returns 0, -1 if the shape is not served, a hipError_t otherwise.  dev_cand sized like the f32 kernel's.
// dev_evals (or NULL): incremented by the number of exact evaluations of the real passes.
extern "C" int vg_batch_h_launch(const uint8_t *dev_rows, int rows_tiled, long long n_rows, long long stride_bytes, int dim, int type_code,
                                 const uint8_t *dev_xrows, long long xstride_bytes,
                                 const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root,
                                 const float *dev_row_nn, uint64_t *dev_cand, int npart, int tiles_per_part,
                                 uint64_t *dev_out_keys, unsigned long long *dev_evals, int waves,
                                 uint64_t *dev_pairs, uint32_t *dev_pair_counts, int pair_cap, hipStream_t stream) {
    const int ntb = vgh_ntb(stride_bytes);
    const bool split = dev_pairs != nullptr && dev_pair_counts != nullptr && pair_cap > 0 && waves == VGH_WAVES_OF(ntb);
    if (!ntb || k < 1 || k > VGH_MAX_K || (waves != VGH_WAVES_OF(ntb) && !(waves == 4 && VGH_HAS_W4(ntb)))) return -1;
    const size_t smem = vgh_lds_bytes(ntb, k, waves);
    const int qpb = waves * VGH_QPW;
    if (smem > (size_t)160 * 1024 || nq_pad % qpb != 0 || npart < 1 || npart > VG_SEL_MAX_HEADS || n_rows < 1) return -1;
    if (mode < VGH_DOT || mode > VGH_L2 || !dev_row_nn) return -1;
    BatchArgsH a;
    a.rows = dev_rows; a.tiled = rows_tiled; a.queries = dev_queries; a.row_nn = dev_row_nn; a.cand = dev_cand;
    a.xrows = dev_xrows; a.xqueries = dev_queries; a.xstride = xstride_bytes;
    a.cerr = (float)(dim + 64) * 4.76837158203125e-7f + (type_code == 2 ? 0.0078125f + 1.52587890625e-5f : 0.0f);   // (D+64) 2^-21 [+ 2u + u^2, u = 2^-8: query AND row are rounded to bf16]
    a.n_rows = n_rows; a.stride = stride_bytes; a.nq_pad = nq_pad; a.nq_real = nq_real; a.npart = npart; a.k = k;
    a.mode = mode; a.root = root; a.dim = dim; a.evals = dev_evals;
    a.pairs = dev_pairs; a.pair_counts = dev_pair_counts; a.pair_cap = pair_cap; a.qnn = nullptr; a.lds_pairs = 0; a.part_group = 0;
    const int G = nq_pad / qpb;
    a.n_regions = G * npart * waves;
    if (split) {
        hipError_t e = hipMemsetAsync(dev_pair_counts + a.n_regions, 0, sizeof(uint32_t), stream);      // the overflow flag
        if (e != hipSuccess) return (int)e;
    }
    const size_t smem_filter = (size_t)VGH_RING_OF(ntb) * ntb * 1024 + 1024 + (size_t)waves * VGH_QPW * (8 + 4 + 4);
    const size_t smem_exact = (size_t)VGH_QPW * (8 + 4 + 4 + 4) + (size_t)VGH_QPW * k * 8;
    auto launch_filter = [&](const BatchArgsH &b) -> int {
        return type_code == 2 ? vgh_launch_filter_f32(&b, ntb, waves, G * ((npart + 7) / 8) * 8, smem_filter, stream)
                              : (type_code == 1 ? vgh_launch_filter_bf16(&b, ntb, waves, G * ((npart + 7) / 8) * 8, smem_filter, stream)
                                                : vgh_launch_filter_f16(&b, ntb, waves, G * ((npart + 7) / 8) * 8, smem_filter, stream));
    };
    auto launch_exact = [&](const BatchArgsH &b) -> int {
        return type_code == 2 ? vgh_launch_exact_f32(&b, ntb, waves, b.n_regions, smem_exact, stream)
                              : (type_code == 1 ? vgh_launch_exact_bf16(&b, ntb, waves, b.n_regions, smem_exact, stream)
                                                : vgh_launch_exact_f16(&b, ntb, waves, b.n_regions, smem_exact, stream));
    };
    const int blocks = G * ((npart + 7) / 8) * 8;
    const long long ntiles = (n_rows + VGH_TILE - 1) / VGH_TILE;
    auto launch = [&](const BatchArgsH &b, bool bound) -> int {
        if (bound) return type_code == 2 ? vgh_launch_bound_f32(&b, ntb, waves, blocks, smem, stream)
                                         : (type_code == 1 ? vgh_launch_bound_bf16(&b, ntb, waves, blocks, smem, stream) : vgh_launch_bound_f16(&b, ntb, waves, blocks, smem, stream));
        if (type_code == 2) return vgh_launch_real_f32(&b, ntb, waves, blocks, smem, stream);
        return type_code == 1 ? vgh_launch_real_bf16(&b, ntb, waves, blocks, smem, stream) : vgh_launch_real_f16(&b, ntb, waves, blocks, smem, stream);
    };
    // Large corpora: a BOUND pre-pass over the first 1/512 of the rows (round 3; 1/32 before) gives every query an upper bound of its final
    // k-th best distance (no exact evaluations - with thresholds starting at +Inf they were a quarter of the whole
    // time); the real pass then scans EVERY row starting from those thresholds.  (1/16 .. 1/32 measured best: 11.7 ms
    // at 1024 x 10M x 384; 1/64 12.2, 1/8 12.4, 1/4 13.9.)
    long long pre = 0;
    {
        const int denom = vg_sw(SW_VG_BATCH_PREPASS, VGB_PREPASS_DENOM_DEFAULT);     // (vg_batch_common.h: re-measured in round 3)
        if (denom > 0 && ntiles >= 65536) pre = ((ntiles / denom + npart - 1) / npart) * npart;      // whole partitions; < 2M rows: one pass
    }
    int rc;
    a.npart_total = npart;                                        // every pass writes (and the merges read) lists 0 .. npart-1
    a.part_base = 0; a.seed = 0;
    if (pre > 0) {
        a.tile_begin = 0; a.tile_end = pre; a.tiles_per_part = (int)(pre / npart); a.init_keys = nullptr;
        if ((rc = launch(a, true)) != 0) return rc;
        if ((rc = vg_batch_merge_launch(dev_cand, nq_pad, a.npart_total, npart, k, dev_out_keys, stream)) != 0) return rc;
        a.init_keys = dev_out_keys;
    } else {
        a.init_keys = nullptr;
    }
    // the real pass, in stages over growing row ranges (vg_batch_common.h): the first one scans the pre-pass rows again (their
    // lists hold bounds, not distances), every later one starts from - and partition 0 carries on - the merged lists so far
    long long bounds[16];
    const int nstages = vgb_stage_bounds(ntiles, pre, bounds, 16, G >= 2 ? 200 : 400);     // (several query groups: doubling)
    for (int s = 0; s < nstages; ++s) {
        a.tile_begin = bounds[s]; a.tile_end = bounds[s + 1];
        a.tiles_per_part = (int)((a.tile_end - a.tile_begin + npart - 1) / npart);
        a.seed = (s > 0) ? 1 : 0;
        if (split) {                                              // filter -> pairs -> exact evaluation + lists
            if ((rc = launch_filter(a)) != 0) return rc;
            if ((rc = launch_exact(a)) != 0) return rc;
        } else if ((rc = launch(a, false)) != 0) return rc;
        if ((rc = vg_batch_merge_launch(dev_cand, nq_pad, a.npart_total, npart, k, dev_out_keys, stream)) != 0) return rc;
        a.init_keys = dev_out_keys;
    }
    return 0;
}
#endif   // VGH_TU
