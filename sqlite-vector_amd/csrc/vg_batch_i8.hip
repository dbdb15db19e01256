// vg_batch_i8.hip - batched queries over a QUANTIZED corpus (uint8 / int8 rows, the vector_quantize format): Q x C^T
// on the integer matrix cores (v_mfma_i32_32x32x32_i8) with the same fused per-query top-k as the f32 kernel
// (vg_batch.hip).  Integer dot products are exact, so every distance is the single-query kernel's distance bit for
// bit (vg_accum.h AccumInt: exact 32-bit sums, one int->float conversion, correctly rounded sqrt / divide) and the
// result lists equal Q separate vector_quantize_scan calls exactly, ties included.
//
//   * a workgroup = 8 wavefronts (two per SIMD) owns 256 queries x one partition of the corpus; A (32 queries per
//     wavefront) is stationary in registers: lane (x, h) keeps bytes [32t + 16h, +16) of query x in a[t] (4 VGPRs per k-step);
//   * B streams through LDS in tiles of 32 rows by LDS-DMA from a TILE-MAJOR copy of the corpus (every DMA instruction moves
//     1 KiB of contiguous memory), transposed by 16-byte chunk - see the kernel; one ds_read_b128 feeds one MFMA;
//   * uint8: the matrix core multiplies SIGNED bytes, so it runs on x' = x - 128 (the copy is XOR 0x80; queries are flipped
//     while they are loaded) and the true dot product is restored exactly:
//         sum q x = sum q'x' + 128 (sum q + sum x) - 16384 L        (L = padded row length, pads are 0 <-> -128)
//     with sum x, sum x^2 per row from cached vectors (vg_i8_rowstat_kernel) and sum q, sum q^2 per query;
//   * gates are integer margins on the raw accumulator for dot / L2 and one multiply per register for cosine; survivors get
//     the exact distance and go through the same list insert as in vg_batch.hip;
//   * the schedule (rows up to 1 KiB) is a software pipeline: a wavefront owns TWO accumulator sets and the gate math of
//     tile t runs under the MFMAs of tile t+1; tiles live in three LDS buffers, the one workgroup barrier per tile sits in
//     the MIDDLE of the k loop (where it only orders buffer reuse), the B-operand reads run on across tile ends, the first
//     wavefront of every SIMD issues the tile's LDS-DMA (contiguous pieces: one M0 set-up, back-to-back instructions);
//   * what round 2 measured about this kernel (timing builds): the two wavefronts of a SIMD run in lockstep, so
//     their non-MFMA work (DMA issue ~90-180 cycles per instruction, tests, barrier) adds to the 1536 MFMA cycles of a tile
//     instead of hiding under them; neither issue priorities, nor a fourth buffer, nor alternating MFMA / everything-else
//     roles with two accumulator chains per wavefront (a lone wavefront then issues an MFMA every 37 cycles, but every role
//     switch costs ~500) changed that.  What did help: fewer survivors (staged passes), the tile-major copy, contiguous DMA;
//   * large corpora: a pre-pass over 1/32 of the rows, then the real pass in stages over doubling row ranges, the lists
//     merged and the thresholds refreshed in between (vg_batch_common.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "vg_batch_common.h"

typedef int vgi_i32x16 __attribute__((ext_vector_type(16)));
typedef int vgi_i32x4 __attribute__((ext_vector_type(4)));

#ifndef VGI_WAVES
#define VGI_WAVES 8                     // two wavefronts per SIMD
#endif
#define VGI_WAVES_LONG 4                // rows of 1 - 2 KiB: A alone is up to 256 registers, one wavefront per SIMD
#define VGI_WAVES_OF(NTB) ((NTB) <= 32 ? VGI_WAVES : VGI_WAVES_LONG)
#define VGI_QPW 32
#define VGI_TILE 32
#define VGI_MAX_K 32
#ifndef VGI_BPIPE
#define VGI_BPIPE 4                     // B-operand register quads in flight (LDS reads issued this many k-steps ahead)
#endif
#ifndef VGI_PIPE
#define VGI_PIPE 1                      // 0: every shape on the barrier-per-tile schedule (A/B measurements)
#endif
// Schedules measured and dropped in rounds 1-2 (profiles/r2b, r2c, r2d, r2g, r2t): the two wavefronts of a SIMD in
// alternating phases (one chain each; two chains each over double tiles), DMA two tiles ahead with counted waits, a 4-buffer
// ring with ready / free counters instead of the barrier (also skewed by half a tile), two accumulator chains over even / odd
// k-steps, B reads 8 k-steps ahead, s_setprio on either wavefront of a SIMD or alternating by half tile.

enum { VGI_DOT = 0, VGI_COS = 1, VGI_L2 = 2 };

#ifndef VGI_ABLATE
#define VGI_ABLATE 0                    // measurement builds of the pipelined schedule (tools/build_i8_variants.sh ab1 -DVGI_ABLATE=1; WRONG results):
#endif                                  //   1 no slow path  2 + no fast test  3 + no DMA (tiles go stale)  4 + no barrier: MFMAs + B reads only
#ifndef VGI_TIMING
#define VGI_TIMING 0                    // measurement builds (tools/tools_i8_timing.py, tools/build_i8_variants.sh timing -DVGI_TIMING=1)
#endif
#if VGI_TIMING
// s_memtime ticks summed over all wavefronts of the REAL passes: whole tile loop | mid-tile wait (DMA landed + barrier) |
// slow path (exact distances + inserts) | first half of the k loop (MFMAs + the previous tile's test) | second half (MFMAs +
// DMA issue) | - | wave-tiles | wave-tiles that entered the slow path
// [8 .. 15]: the mid-tile wait per wavefront index 0 .. 7 (the wavefront that waits least is the one the others wait for)
__device__ unsigned long long vgi_ticks[16];
extern "C" int vg_batch_i8_timing(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(vgi_ticks), sizeof(vgi_ticks)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vgi_ticks), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define VGI_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define VGI_TICK(var)
#endif

struct BatchArgsI8 {
    const uint8_t *rows;      // the TILE-MAJOR copy in SIGNED representation (int8 bytes as they are, uint8 XOR 0x80): tile t = rows
                              // 32t .. 32t+31 = 32 * stride contiguous bytes, chunk column c of the 32 rows at c * 512 + row * 16
    const uint8_t *queries;   // nq_pad x stride bytes in the corpus' own (unflipped) representation, zero padded
    const uint32_t *row_stat; // per row: sum x (original representation, int32 bits), sum x^2
    uint64_t *cand;
    long long n_rows;
    long long stride;         // bytes per row (multiple of 16)
    int nq_pad, nq_real, npart, k;   // queries [nq_real, nq_pad) are padding
    int mode, root, is_u8;
    int tiles_per_part;
    long long tile_begin, tile_end;
    int part_base, npart_total;
    const uint64_t *init_keys;
    int seed;                 // staged real passes (vg_batch_common.h): partition 0 starts its lists from init_keys
};

typedef float vgi_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned vgi_u32x2 __attribute__((ext_vector_type(2)));

template <int CTRL> __device__ __forceinline__ uint64_t vgi_dpp64(uint64_t v) {
    return ((uint64_t)vg_dpp_u32<CTRL>((uint32_t)(v >> 32)) << 32) | vg_dpp_u32<CTRL>((uint32_t)v);
}
__device__ __forceinline__ uint64_t vgi_min64(uint64_t a, uint64_t b) { return a < b ? a : b; }

template <int OFF>
__device__ __forceinline__ void vgi_lds_read128(vgi_i32x4 &dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void vgi_wait_lds(vgi_i32x4 &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
// tile buffers of the pipelined schedule (the DMA runs one tile ahead of the reads; a fourth buffer and DMA two tiles ahead
// with counted vmcnt waits was measured: the mid-tile wait did not shrink - it is not the DMA's latency)
#define VGI_PIPE_NBUF 3

// NTB = 32-byte k-steps per row (rows up to NTB * 32 bytes)
// PRE = the pre-pass variant (large corpora): only the SMALLEST distance of each query in a tile enters its list.  k
// entries then stand for k different rows, so the k-th of them bounds the query's final k-th best distance from above:
// the start threshold of the real pass.  A list warms up with one insert per (query, tile) instead of one per passing row.
// PIPE = the software-pipelined schedule (NTB <= 32).
template <int NTB, int MODE, bool IS_U8, bool PRE, bool PIPE>
__global__ __launch_bounds__(64 * VGI_WAVES_OF(NTB), 1) void vg_batch_i8_kernel(BatchArgsI8 a) {
    constexpr int WAVES = VGI_WAVES_OF(NTB), THREADS = 64 * WAVES, QPB = WAVES * VGI_QPW;
    constexpr bool COS = (MODE == VGI_COS), L2M = (MODE == VGI_L2);
    static_assert(!PIPE || NTB <= 32, "the pipelined schedule keeps two accumulator sets next to A: rows up to 1 KiB");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // LDS tile, TRANSPOSED BY 16-BYTE CHUNK: chunk column c of the 32 rows is one contiguous 512-byte run
    // (address c * 512 + row * 16).  A lane's b128 read of (row x, chunk 2t + h) sits next to its neighbours' - no
    // bank conflicts, no row padding - and one LDS-DMA instruction (64 lanes x 16 bytes = two chunk columns) carries
    // 64 (row, chunk) pairs whatever the row length.
    constexpr int TILE_BYTES = NTB * 2 * 512;
    constexpr int L = NTB * 32;                                 // padded row length the matrix core sees
    constexpr int NBUF = PIPE ? VGI_PIPE_NBUF : 2;
    uint8_t *tile0 = smem;
    uint32_t *rstat_lds = reinterpret_cast<uint32_t *>(smem + NBUF * TILE_BYTES);        // 2 slots x 4 tiles x [32 rows][sum x, sum x^2]
    uint32_t *qstat_lds = rstat_lds + 2 * 256;                                           // [waves][32][2]: sum q, sum q^2
    float *thr_lds = reinterpret_cast<float *>(qstat_lds + WAVES * VGI_QPW * 2);         // [waves][32]: current k-th best distance
    uint64_t *lists = reinterpret_cast<uint64_t *>(thr_lds + WAVES * VGI_QPW);           // [waves][32][k]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31, h = lane >> 5;
    const int k = a.k;

    const int G = a.nq_pad / QPB;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int g = idx % G;
    const int part = (idx / G) * 8 + xcd;
    if (part >= a.npart) return;
    const int q0 = g * QPB + wave * VGI_QPW;

    // ---- A operand (+ the query's sums in its ORIGINAL representation)
    vgi_i32x4 areg[NTB];
    uint32_t sq_part = 0, sqq_part = 0;
    {
        const uint8_t *qrow = a.queries + (long long)(q0 + x) * a.stride;
        vgb_static_for<0, NTB>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int off = 32 * t + 16 * h;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (off < a.stride) v = *reinterpret_cast<const uint4 *>(qrow + off);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (IS_U8) {
                    sq_part = __builtin_amdgcn_udot4(w[j], 0x01010101u, sq_part, false);
                    sqq_part = __builtin_amdgcn_udot4(w[j], w[j], sqq_part, false);
                } else {
                    sq_part = (uint32_t)__builtin_amdgcn_sdot4((int)w[j], 0x01010101, (int)sq_part, false);
                    sqq_part = (uint32_t)__builtin_amdgcn_sdot4((int)w[j], (int)w[j], (int)sqq_part, false);
                }
            }
            const uint32_t flip = IS_U8 ? 0x80808080u : 0u;
            areg[t] = vgi_i32x4{(int)(v.x ^ flip), (int)(v.y ^ flip), (int)(v.z ^ flip), (int)(v.w ^ flip)};
        });
    }
    uint32_t *qs_w = qstat_lds + wave * VGI_QPW * 2;
    float *thr_w = thr_lds + wave * VGI_QPW;
    uint64_t *wave_lists = lists + (size_t)wave * VGI_QPW * k;
    {
        const uint32_t sq = sq_part + __shfl_xor(sq_part, 32), sqq = sqq_part + __shfl_xor(sqq_part, 32);
        if (h == 0) {
            qs_w[2 * x] = sq; qs_w[2 * x + 1] = sqq;
            // the pre-pass bound is the distance of an actual row, which the first real stage meets again: one ulp up, so
            // that the strict comparison of the slow path lets a row AT the bound in; padding queries never accept
            float t = a.init_keys ? nextafterf(vgb_kth_distance(a.init_keys[(long long)(q0 + x) * 64 + (k - 1)]), INFINITY) : INFINITY;
            if (q0 + x >= a.nq_real) t = -INFINITY;
            thr_w[x] = t;
        }
        const bool seeded = a.seed != 0 && part == 0;               // (keys of rows no later stage meets again)
        for (int s = lane; s < VGI_QPW * k; s += 64)
            wave_lists[s] = seeded ? a.init_keys[(long long)(q0 + s / k) * 64 + s % k] : VG_EMPTY_KEY;
    }
    // pad columns never touched by the DMA must read as "0" of the original representation
    for (int s = tid; s < NBUF * TILE_BYTES / 4; s += THREADS) reinterpret_cast<uint32_t *>(tile0)[s] = IS_U8 ? 0x80808080u : 0u;
    __syncthreads();

    // ---- tile streaming by LDS-DMA: piece p = chunk columns 2p and 2p+1 of all 32 rows = 1 KiB of the tile-major copy;
    // wavefront w moves pieces w, w + WAVES, ...  Lane l's 16 bytes land at M0 + 16 * l.
    const int chunks_per_row = (int)(a.stride / 16);
    const int npieces = (chunks_per_row + 1) / 2;               // pieces that carry data (<= NTB)
    const long long tile_first = a.tile_begin + (long long)part * a.tiles_per_part;
    const long long tile_last = min(tile_first + a.tiles_per_part, a.tile_end);
    const unsigned long long stride_b = (unsigned long long)a.stride;
    const uint32_t lds_tile0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)tile0;
#ifndef VGI_DMA_G0
#define VGI_DMA_G0 1                    // the first wavefront of every SIMD (waves 0-3: the arbiter favours the older wavefront, they reach the
                                        // barrier ~400 cycles early whatever they do) moves ALL pieces; 0: every wavefront its share (+2 %)
#endif
    constexpr int NISSUE = (VGI_DMA_G0 && WAVES == 8) ? 4 : WAVES;    // wavefronts that issue DMA
    constexpr int NPIECE = (NTB + NISSUE - 1) / NISSUE;               // piece slots per issuing wavefront and tile (compile time)
    // issuing wavefront w moves the CONTIGUOUS pieces w * NPIECE .. w * NPIECE + NPIECE - 1: one M0 set-up serves up to four
    // of them (the instruction offset moves the global and the LDS address alike)
    uint64_t piece_mask[NPIECE];
    bool all_full = wave < NISSUE;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int p = wave * NPIECE + i;
        piece_mask[i] = __ballot(wave < NISSUE && p < npieces && (2 * p + h) < chunks_per_row);
        all_full = all_full && piece_mask[i] == ~0ull;
    }
    // (the tile-major copy holds whole tiles: rows past the end of the corpus read whatever the allocation holds there,
    //  their scores are masked by the row bound)
    const uint32_t lane_goff = (uint32_t)lane * 16u;
    auto dma_piece = [&](long long tile, int buf, int i) {
        if (piece_mask[i] == 0) return;
        const int p = wave * NPIECE + i;
        const uint8_t *sbase = a.rows + (unsigned long long)(tile * VGI_TILE) * stride_b + (unsigned)p * 1024u;   // 1 KiB contiguous
        const uint32_t lds_dst = lds_tile0 + (uint32_t)(buf * TILE_BYTES + p * 1024);
        uint32_t keep;
        uint64_t keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_and_b64 exec, exec, %5\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keep_exec) : "v"(lane_goff), "s"(sbase), "s"(lds_dst), "s"(piece_mask[i]) : "memory", "scc");
    };
    // N (1 .. 4) whole pieces starting at piece slot i0, back to back
    auto dma_run = [&](long long tile, int buf, auto i0c, auto nc) {
        constexpr int i0 = decltype(i0c)::value, N = decltype(nc)::value;
        const uint8_t *sbase = a.rows + (unsigned long long)(tile * VGI_TILE) * stride_b + (unsigned)(wave * NPIECE + i0) * 1024u;
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
    auto dma_tile = [&](long long tile, int buf) {
        if (all_full) {                                              // the common case: whole pieces, back to back in runs of <= 4
            vgb_static_for<0, (NPIECE + 3) / 4>([&](auto rc) {
                constexpr int i0 = 4 * decltype(rc)::value, N = NPIECE - i0 < 4 ? NPIECE - i0 : 4;
                dma_run(tile, buf, std::integral_constant<int, i0>{}, std::integral_constant<int, N>{});
            });
            return;
        }
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) dma_piece(tile, buf, pc);
    };
    // The per-row sums (sum x, sum x^2: 8 bytes per row) ride the same pipeline, FOUR tiles per LDS-DMA instruction (1 KiB) into a
    // two-slot ring; the wavefronts take turns issuing it.  (One instruction per tile from the same wavefront made that
    // wavefront ~360 cycles late at every barrier; loads into registers that stay in flight across the loop's back edge are not
    // safe from compiler-inserted copies; compiler-placed loads cost a full L2 / HBM round trip per tile.)
    const uint32_t lds_rstat0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)rstat_lds;
    auto dma_stat_group = [&](long long tile4, int slot) {          // tiles tile4 .. tile4 + 3
        const uint8_t *sbase = reinterpret_cast<const uint8_t *>(a.row_stat + tile4 * (VGI_TILE * 2));
        const uint32_t lds_dst = lds_rstat0 + (uint32_t)(slot * 1024);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_goff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    const uint32_t stat_goff = (uint32_t)x * 8u;
    auto load_stats = [&](long long tile, vgi_u32x2 &dst) {          // (the barrier-per-tile schedule: straight into registers)
        const uint32_t *p = a.row_stat + tile * (VGI_TILE * 2);
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(stat_goff), "s"(p) : "memory");
    };

    // ---- what the tile boundary tests.  Register r of lane (x, h) belongs to query qi(r, h) = (r&3) + 8*(r>>2) + 4*h.
    // The FAST path keeps one value per register - everything else a survivor needs (the query's sums, its current k-th
    // best distance) stays in LDS and is read on the slow path only:
    //   dot     ng = -(floor(-thr) - 8 - cq)                  a pair can pass  <=  acc + ng + cx >= 0
    //   L2      ng = -(qq - 2 cq - ceil(thr2 (1 + 1e-6)) - 8)                  <=  2 acc + ng - (xx - 2 cx) >= 0
    //   cosine  invg = (1 + 1e-6) / G,  G = (1 - thr) |q| (1 - 1e-5 sgn) > 0: the accumulator STARTS at cq + 1 (integer
    //           sums: exact), so acc + cx = q.x + 1 and  q.x >= G |x| - 1  <=>  (q.x + 1) / G >= |x|; registers whose gate
    //           is still open (G <= 0: list not full) are flagged in open_mask and always pass
    // with cq / cx the query / row parts of what turns the raw accumulator into sum q x (uint8: 128 sum q - 16384 L, 128 sum x).
    // The exact test - (float) distance < thr, the single-query kernel's arithmetic - is the slow path's.
    // (Vectors, not arrays: the slow path indexes them with a run-time register number.)
    vgi_i32x16 ng, cinit;
    vgi_f32x16 invg;
    uint32_t open_mask = 0;
    const bool l2_root = a.root != 0;
    auto as_float_like = [](uint32_t v) -> float { return IS_U8 ? (float)v : (float)(int32_t)v; };
    struct Gate { int ng; float invg; bool open; };
    auto gate_of = [&](uint32_t sq, uint32_t sqq, float thr) __attribute__((always_inline)) -> Gate {
        const int cq = IS_U8 ? (int)(128u * sq) - 16384 * L : 0;
        Gate gt;
        gt.ng = -1500000000; gt.invg = 0.0f; gt.open = false;
        if (thr == -INFINITY) return gt;                       // padding (an all-zero query ties every row at cosine 1.0)
        if (COS) {
            const float Gf = (1.0f - thr) * sqrtf(as_float_like(sqq));
            const float gate_f = fmaxf(Gf - 1e-5f * fabsf(Gf), -3.0e38f);
            const bool closed = gate_f > 1e-30f;
            gt.invg = closed ? __fdividef(1.0f + 1e-6f, gate_f) : 0.0f;
            gt.open = !closed;
        } else if (L2M) {
            const float thr2 = l2_root ? thr * thr : thr;
            const float Tf = thr2 * (1.0f + 1e-6f) + 8.0f;
            const int T = (Tf < 1.5e9f) ? (int)Tf : 1500000000;           // +Inf / NaN: accept everything (totals < 2^27)
            gt.ng = T + 2 * cq - (int)sqq;
        } else {
            const float Gf = -thr - 8.0f;
            const int Gq = (Gf > -1.5e9f) ? (int)floorf(Gf) : -1500000000;  // thr = +Inf: accept everything (|qx| < 2^27)
            gt.ng = cq - Gq;
        }
        return gt;
    };
    vgb_static_for<0, 16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const uint32_t sq = qs_w[2 * qi], sqq = qs_w[2 * qi + 1];
        const Gate gt = gate_of(sq, sqq, thr_w[qi]);
        ng[r] = gt.ng; invg[r] = gt.invg;
        cinit[r] = COS ? (IS_U8 ? (int)(128u * sq) - 16384 * L : 0) + 1 : 0;
        open_mask |= gt.open ? (1u << r) : 0u;
    });

    // the fast test of a tile, in 8 pieces of two registers each (the pipelined schedule spreads them over the k loop of the
    // NEXT tile); jm accumulates the lane's best margin
    auto judge_item = [&](auto ic, const vgi_i32x16 &acc, int cx, int &jm_i, float &jm_f) __attribute__((always_inline)) {
        constexpr int r0 = 2 * decltype(ic)::value, r1 = r0 + 1;
        if constexpr (COS) {
            const float f0 = as_float_like((uint32_t)(acc[r0] + cx)) * invg[r0], f1 = as_float_like((uint32_t)(acc[r1] + cx)) * invg[r1];
            jm_f = fmaxf(jm_f, fmaxf(f0, f1));
        } else {
            const int m0 = (L2M ? 2 * acc[r0] : acc[r0]) + ng[r0], m1 = (L2M ? 2 * acc[r1] : acc[r1]) + ng[r1];
            const int mm = m0 > m1 ? m0 : m1;
            jm_i = jm_i > mm ? jm_i : mm;
        }
    };
    auto judge_any = [&](int jm_i, float jm_f, int cx, uint32_t xx) -> bool {
        if constexpr (COS) return __ballot(open_mask != 0u || jm_f >= sqrtf(as_float_like(xx))) != 0;
        else return __ballot(jm_i - (L2M ? (int)xx - 2 * cx : -cx) >= 0) != 0;
    };

    // the slow path of a tile: which registers hold a passing pair (16 ballots), then - ONE copy of the code, looping over the
    // pending registers by run-time index - the exact distance of each pair from the raw accumulator (the single-query kernel's
    // epilogue, vg_accum.h), the list inserts, the refreshed gate.  (Unrolled over the 16 registers, the slow paths were most
    // of the kernel's code: entering one cost ~16k cycles of instruction-cache misses.)
    auto judge_slow = [&](const vgi_i32x16 &acc, long long tile, int sx, uint32_t xx) __attribute__((always_inline)) {
        const long long row = tile * VGI_TILE + x;
        const int cx = IS_U8 ? 128 * sx : 0;
        unsigned pend = 0;
        if constexpr (COS) {
            const float nb = sqrtf(as_float_like(xx));
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const bool p = ((open_mask >> r) & 1u) || as_float_like((uint32_t)(acc[r] + cx)) * invg[r] >= nb;
                pend |= __ballot(p) ? (1u << r) : 0u;
            });
        } else {
            const int hx = L2M ? (int)xx - 2 * cx : -cx;
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                pend |= __ballot((L2M ? 2 * acc[r] : acc[r]) + ng[r] - hx >= 0) ? (1u << r) : 0u;
            });
        }
        while (pend) {
            const int r = __builtin_ctz(pend);
            pend &= pend - 1;
            const int q_lo = (r & 3) + 8 * (r >> 2), qi = q_lo + 4 * h;
            const uint32_t sq = qs_w[2 * qi], sqq = qs_w[2 * qi + 1];
            const int cq = IS_U8 ? (int)(128u * sq) - 16384 * L : 0;
            float thr_l = thr_w[qi];
            const int raw = acc[r] - cinit[r];                              // the raw accumulator
            const uint32_t qx = (uint32_t)(raw + cq + cx);                  // sum q x, modulo 2^32 like the reference
            float d;
            if (COS) {
                d = vg_cosine_from_norms(as_float_like(qx), sqrtf(as_float_like(sqq)), sqrtf(as_float_like(xx)));
            } else if (L2M) {
                const float t = (float)(uint32_t)(sqq + xx - 2u * qx);
                d = l2_root ? sqrtf(t) : t;
            } else {
                d = -as_float_like(qx);
            }
            d = vg_clamp(d);
            // strict: rows arrive in scan order, a row that only ties the k-th best has the larger position and loses
            const bool pass = (row < a.n_rows) && (d < thr_l);
            unsigned long long m = __ballot(pass);
            if (!m) continue;
            if constexpr (PRE) {
                uint64_t kmin = pass ? vg_make_key(d, (uint32_t)row) : VG_EMPTY_KEY;     // min over the 32 rows of each query
                kmin = vgi_min64(kmin, vgi_dpp64<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(kmin));
                kmin = vgi_min64(kmin, vgi_dpp64<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(kmin));
                kmin = vgi_min64(kmin, vgi_dpp64<VG_DPP_ROW_HALF_MIRROR>(kmin));
                kmin = vgi_min64(kmin, vgi_dpp64<VG_DPP_ROW_MIRROR>(kmin));
                kmin = vgi_min64(kmin, (uint64_t)__shfl_xor((unsigned long long)kmin, 16));
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint64_t c = vg_readlane64(kmin, 32 * hh);
                    if (c == VG_EMPTY_KEY) continue;
                    const float nt = vgb_kth_distance(vgb_list_insert(wave_lists + (q_lo + 4 * hh) * k, k, lane, c));
                    if (h == hh) thr_l = fminf(nt, thr_l);
                }
            } else {
                const uint64_t key = vg_make_key(d, (uint32_t)row);
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int hh = src >> 5;
                    const uint64_t c = vg_readlane64(key, src);
                    const float nt = vgb_kth_distance(vgb_list_insert(wave_lists + (q_lo + 4 * hh) * k, k, lane, c));
                    if (h == hh) thr_l = fminf(nt, thr_l);
                }
            }
            if (x == 0) thr_w[qi] = thr_l;                                  // (the 32 lanes of a half hold the same value)
            const Gate gt = gate_of(sq, sqq, thr_l);
            ng[r] = gt.ng; invg[r] = gt.invg;
            open_mask = gt.open ? (open_mask | (1u << r)) : (open_mask & ~(1u << r));
        }
    };
    // fast test + slow path in one piece (the barrier-per-tile schedule, and the last tile of the pipelined one)
    auto judge_all = [&](const vgi_i32x16 &acc, long long tile, vgi_u32x2 st) __attribute__((always_inline)) {
        const int sx = (int)st[0];
        const uint32_t xx = st[1];
        const int cx = IS_U8 ? 128 * sx : 0;
        int jm_i = -0x7FFFFFFF;
        float jm_f = -INFINITY;
        vgb_static_for<0, 8>([&](auto ic) { judge_item(ic, acc, cx, jm_i, jm_f); });
        if (judge_any(jm_i, jm_f, cx, xx)) judge_slow(acc, tile, sx, xx);
    };

    constexpr int BP = VGI_BPIPE < NTB / 2 ? VGI_BPIPE : NTB / 2;
    vgi_i32x4 bq[BP];
    const int T = (int)(tile_last - tile_first);
#if VGI_TIMING
    unsigned long long tk_sync = 0, tk_slow = 0, tk_nslow = 0, tk_h1 = 0, tk_h2 = 0, tk_mid = 0;
#endif

    if constexpr (PIPE) {
        // SOFTWARE PIPELINE.  Tile t lives in buffer t % 3.  One step = the k loop of tile t into accumulator set `cur`, with, in
        // between its MFMAs,
        //   k-steps 1 .. M-1   the fast test of tile t-1 (accumulator set `prev`: its last MFMA was issued one k-step ago, so
        //                      neither the MFMA drain nor the gate math ever stops the matrix pipe),
        //   k-step  M          THE tile's synchronisation: my DMA pieces of tile t+1 (issued during tile t-1) have landed,
        //                      barrier - from here on tile t+1 is readable and nobody reads tile t-1 any more - then the slow
        //                      path of tile t-1 if a pair passed (its wavefront falls behind; the others wait at the next
        //                      barrier),
        //   k-step  M+1        the DMA of tile t+2 into the buffer of tile t-1 (by the wavefronts that issue DMA: VGI_DMA_G0),
        //   the last BP steps  the first B reads of tile t+1: the read pipeline runs on across the tile end.
        constexpr int M = NTB / 2;
        static_assert(BP <= NTB - M && 2 <= NTB - M - 1, "pipeline shape");
        const int grp = wave >= WAVES / 2 ? 1 : 0;
        if (T > 0) {
            dma_tile(tile_first, 0);
            dma_tile(min(tile_first + 1, tile_last - 1), 1);
            if (wave == 0) dma_stat_group(tile_first, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (T > 0) {
            vgi_i32x16 accA, accB;
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[r] = 0;
            const uint32_t lane_b = lds_tile0 + (uint32_t)(h * 512 + x * 16);
            const uint32_t lane_stat = lds_rstat0 + (uint32_t)(x * 8);
            vgb_static_for<0, BP>([&](auto tc) { vgi_lds_read128<1024 * decltype(tc)::value>(bq[decltype(tc)::value], lane_b); });
            VGI_TICK(t_loop0);
            int bcur = 0;                                                    // buffer of tile t
            // `car` = the row sums of the tile under test: read (inline asm, like the B operand) just before a step's
            // synchronisation, used by the next step
            vgi_u32x2 car = {0u, 0u};
            auto step = [&](vgi_i32x16 &cur, const vgi_i32x16 &prev, int ti, bool prev_valid) __attribute__((always_inline)) {
                const int bnext = bcur + 1 == NBUF ? 0 : bcur + 1, bdma = bcur == 0 ? NBUF - 1 : bcur - 1;
                const long long tile = tile_first + ti;
                const long long tile_dma = min(tile + 2, tile_last - 1);
                const uint32_t base_cur = lane_b + (uint32_t)(bcur * TILE_BYTES), base_next = lane_b + (uint32_t)(bnext * TILE_BYTES);
                const int psx = (int)car[0];                                 // tile t-1
                const uint32_t pxx = car[1];
                const int pcx = IS_U8 ? 128 * psx : 0;
                // the row sums of tiles t+2 .. t+5 go with tile t+2's pieces when t+2 starts a group of four
                const bool stat_turn = ((ti + 2) & 3) == 0 && wave == (((ti + 2) >> 2) & (NISSUE - 1));
                int jm_i = -0x7FFFFFFF;
                float jm_f = -INFINITY;
                VGI_TICK(tstep0);
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[r] = cinit[r];
                vgb_static_for<0, NTB>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    vgi_wait_lds<BP - 1>(bq[t % BP]);
                    cur = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[t], bq[t % BP], cur, 0, 0, 0);
                    if constexpr (t == M) {
                        const bool any = VGI_ABLATE >= 1 ? false : (judge_any(jm_i, jm_f, pcx, pxx) && prev_valid);
                        if (VGI_ABLATE >= 2) asm volatile("" :: "v"(prev));        // (the accumulators stay alive without their test)
                        VGI_TICK(ts0);
                        // this tile's row sums for the next step; then every LDS read in flight has returned (the slow path may
                        // move registers around), my DMA pieces of tile t+1 have landed, barrier
                        const uint32_t sa = lane_stat + (uint32_t)((((ti >> 2) & 1) << 10) + ((ti & 3) << 8));
                        asm volatile("ds_read_b64 %0, %1" : "=v"(car) : "v"(sa) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(car));
                        vgb_static_for<0, BP>([&](auto bc) { vgi_wait_lds<0>(bq[decltype(bc)::value]); });
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (VGI_ABLATE < 4) __syncthreads();
                        VGI_TICK(ts1);
                        if (any) judge_slow(prev, tile - 1, psx, pxx);
#if VGI_TIMING
                        const unsigned long long ts2 = __builtin_readcyclecounter();
                        tk_sync += ts1 - ts0; tk_slow += ts2 - ts1; tk_nslow += any ? 1 : 0; tk_h1 += ts0 - tstep0; tk_mid = ts2;
#endif
                    }
                    if constexpr (t + BP < NTB) vgi_lds_read128<1024 * (t + BP)>(bq[t % BP], base_cur);
                    else vgi_lds_read128<1024 * (t + BP - NTB)>(bq[t % BP], base_next);
                    if constexpr (VGI_ABLATE < 2 && t >= 1 && t < M) {
                        vgb_static_for<0, 8>([&](auto ic) {
                            if constexpr (1 + decltype(ic)::value * (M - 1) / 8 == t) {
                                judge_item(ic, prev, pcx, jm_i, jm_f);
                                // (pinned to this k-step: left alone, the optimizer sinks the whole test next to its use)
                                if constexpr (COS) asm volatile("" : "+v"(jm_f)); else asm volatile("" : "+v"(jm_i));
                            }
                        });
                    }
                    if constexpr (VGI_ABLATE < 3 && t == M + 1) { if (grp == 0) dma_tile(tile_dma, bdma); }
                    if constexpr (VGI_ABLATE < 3 && t == M + 1 + (NTB - M - 1) / 2) { if (grp == 1 && NISSUE == WAVES) dma_tile(tile_dma, bdma); }
                    if constexpr (VGI_ABLATE < 3 && t == NTB - 1) { if (stat_turn) dma_stat_group(tile + 2, ((ti + 2) >> 2) & 1); }
                    __builtin_amdgcn_sched_barrier(0);
                });
#if VGI_TIMING
                tk_h2 += __builtin_readcyclecounter() - tk_mid;
#endif
                bcur = bnext;
            };
            for (int ti = 0; ti < T; ti += 2) {
                step(accA, accB, ti, ti > 0);
                if (ti + 1 < T) step(accB, accA, ti + 1, true);
            }
            // the last tile: nothing left to hide its test under (`car` holds its row sums since its own mid-step)
            vgb_static_for<0, BP>([&](auto bc) { vgi_wait_lds<0>(bq[decltype(bc)::value]); });
            if (VGI_ABLATE >= 1) { asm volatile("" :: "v"(accA)); asm volatile("" :: "v"(accB)); }
            else if (T & 1) judge_all(accA, tile_last - 1, car);
            else judge_all(accB, tile_last - 1, car);
#if VGI_TIMING
            if (!PRE && lane == 0) {
                const unsigned long long t_end = __builtin_readcyclecounter();
                atomicAdd(&vgi_ticks[0], t_end - t_loop0); atomicAdd(&vgi_ticks[1], tk_sync); atomicAdd(&vgi_ticks[2], tk_slow);
                atomicAdd(&vgi_ticks[3], tk_h1); atomicAdd(&vgi_ticks[4], tk_h2);
                atomicAdd(&vgi_ticks[6], (unsigned long long)T); atomicAdd(&vgi_ticks[7], tk_nslow);
                atomicAdd(&vgi_ticks[8 + (wave & 7)], tk_sync);
            }
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // BARRIER PER TILE (rows of 1 - 2 KiB, and long lists next to 1 KiB rows: no room for a second accumulator set / a
        // third tile buffer): k loop with the DMA of the next tile, test, barrier
        vgi_i32x16 acc;
        vgi_u32x2 st = {0u, 0u};
        if (T > 0) dma_tile(tile_first, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (long long tile = tile_first; tile < tile_last; ++tile) {
            const int cur_buf = (int)((tile - tile_first) & 1), next_buf = cur_buf ^ 1;
            const long long tile_next = min(tile + 1, tile_last - 1);      // (the last iteration re-fetches its own tile)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = cinit[r];
            const uint32_t baddr = lds_tile0 + (uint32_t)(cur_buf * TILE_BYTES + h * 512 + x * 16);
            vgb_static_for<0, BP>([&](auto tc) { vgi_lds_read128<1024 * decltype(tc)::value>(bq[decltype(tc)::value], baddr); });
            vgb_static_for<0, NTB>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int in_flight_after = (NTB - 1 - t) < (BP - 1) ? (NTB - 1 - t) : (BP - 1);
                vgi_wait_lds<in_flight_after>(bq[t % BP]);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[t], bq[t % BP], acc, 0, 0, 0);
                if constexpr (t + BP < NTB) vgi_lds_read128<1024 * (t + BP)>(bq[t % BP], baddr);
                // this tile's row sums and the DMA pieces of the next tile, spread over the first half of the k loop
                if constexpr (t == 0) load_stats(tile, st);
                constexpr int NTD = (NTB + 1) / 2;
                constexpr int pc_lo = (t >= NTD) ? NPIECE : (t * NPIECE + NTD - 1) / NTD;
                constexpr int pc_hi = (t >= NTD) ? NPIECE : (t + 1 == NTD ? NPIECE : ((t + 1) * NPIECE + NTD - 1) / NTD);
                vgb_static_for<pc_lo, pc_hi>([&](auto pcc) { dma_piece(tile_next, next_buf, decltype(pcc)::value); });
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(st) :: "memory");
            judge_all(acc, tile, st);
            __syncthreads();
        }
    }

    for (int s = lane; s < VGI_QPW * 64; s += 64) {
        const int qi = s >> 6, slot = s & 63;
        a.cand[((long long)(q0 + qi) * a.npart_total + a.part_base + part) * 64 + slot] = (slot < k) ? wave_lists[qi * k + slot] : VG_EMPTY_KEY;
    }
}

#ifndef VGI_TU_PRE
// ---- per-row sums of the ORIGINAL representation + the tile-major copy the matrix core reads (uint8: XOR 0x80)
template <bool IS_U8>
__global__ __launch_bounds__(256) void vg_i8_rowstat_kernel(const uint8_t *rows, long long row0, long long n, long long stride,
                                                            uint32_t *stat, uint8_t *tiled) {
    const int sub = threadIdx.x & 15;
    const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ngroups = ((long long)gridDim.x * blockDim.x) >> 4;
    const int nch = (int)(stride / 16);
    for (long long r = group; r < n; r += ngroups) {
        const uint4 *p = reinterpret_cast<const uint4 *>(rows + (row0 + r) * stride);
        // chunk c of row R lives at tile (R / 32) * (32 * stride) + c * 512 + (R % 32) * 16 of the tile-major copy
        const long long R = row0 + r;
        uint8_t *o = tiled ? tiled + (R >> 5) * (32 * stride) + (R & 31) * 16 : nullptr;
        const uint32_t flip = IS_U8 ? 0x80808080u : 0u;
        uint32_t s1 = 0, s2 = 0;
        for (int c = sub; c < nch; c += 16) {
            const uint4 v = p[c];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (IS_U8) { s1 = __builtin_amdgcn_udot4(w[j], 0x01010101u, s1, false); s2 = __builtin_amdgcn_udot4(w[j], w[j], s2, false); }
                else { s1 = (uint32_t)__builtin_amdgcn_sdot4((int)w[j], 0x01010101, (int)s1, false); s2 = (uint32_t)__builtin_amdgcn_sdot4((int)w[j], (int)w[j], (int)s2, false); }
            }
            if (o) *reinterpret_cast<uint4 *>(o + (long long)c * 512) = make_uint4(v.x ^ flip, v.y ^ flip, v.z ^ flip, v.w ^ flip);
        }
        s1 += __shfl_xor(s1, 8); s1 += __shfl_xor(s1, 4); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 1);
        s2 += __shfl_xor(s2, 8); s2 += __shfl_xor(s2, 4); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 1);
        if (sub == 0) { stat[2 * (row0 + r)] = s1; stat[2 * (row0 + r) + 1] = s2; }
    }
}

extern "C" int vg_i8_rowstat_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int is_u8,
                                    uint32_t *dev_stat, uint8_t *dev_flipped, hipStream_t stream) {
    if (n <= 0) return 0;
    long long blocks = (n * 16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (is_u8) hipLaunchKernelGGL((vg_i8_rowstat_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride, dev_stat, dev_flipped);
    else hipLaunchKernelGGL((vg_i8_rowstat_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride, dev_stat, dev_flipped);
    return (int)hipGetLastError();
}

#endif   // !VGI_TU_PRE

// ---- host side
// The kernel instantiations are split over two translation units compiled from this file (build.py): the real-pass
// kernels here, the PRE kernels with -DVGI_TU_PRE (vg_batch_i8_pre.o); -DVGI_TU_ALL: both in this one unit (the
// measurement builds of tools/build_i8_variants.sh).
extern "C" int vgi_launch_pre(const BatchArgsI8 *a, int ntb, int blocks, size_t smem, hipStream_t stream);
extern "C" int vgi_launch_real(const BatchArgsI8 *a, int ntb, int blocks, size_t smem, hipStream_t stream);

// LDS bytes of one workgroup: tile buffers (+ the tiles' row sums), the queries' sums and thresholds, the lists
static size_t vgi_lds(int NTB, int k, int nbuf) {
    const size_t waves = (size_t)VGI_WAVES_OF(NTB);
    return (size_t)nbuf * (NTB * 1024) + 2 * 1024 + waves * VGI_QPW * 2 * 4 + waves * VGI_QPW * 4 + waves * VGI_QPW * k * 8;
}
// the pipelined schedule: rows up to 1 KiB whose tile buffers fit next to the lists (1 KiB rows: k <= 30)
static bool vgi_pipelined(int NTB, int k) { return VGI_PIPE && NTB <= 32 && vgi_lds(NTB, k, VGI_PIPE_NBUF) <= 160 * 1024; }

template <int NTB, int MODE, bool IS_U8, bool PRE, bool PIPE>
static int launch_i8(const BatchArgsI8 &a, int blocks, size_t smem, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vg_batch_i8_kernel<NTB, MODE, IS_U8, PRE, PIPE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((vg_batch_i8_kernel<NTB, MODE, IS_U8, PRE, PIPE>), dim3((unsigned)blocks), dim3(64 * VGI_WAVES_OF(NTB)), smem, stream, a);
    return (int)hipGetLastError();
}
template <int NTB, bool PRE, bool PIPE>
static int launch_i8_mode(const BatchArgsI8 &a, int blocks, size_t smem, hipStream_t stream) {
    if (a.mode == VGI_COS) return a.is_u8 ? launch_i8<NTB, VGI_COS, true, PRE, PIPE>(a, blocks, smem, stream) : launch_i8<NTB, VGI_COS, false, PRE, PIPE>(a, blocks, smem, stream);
    if (a.mode == VGI_L2) return a.is_u8 ? launch_i8<NTB, VGI_L2, true, PRE, PIPE>(a, blocks, smem, stream) : launch_i8<NTB, VGI_L2, false, PRE, PIPE>(a, blocks, smem, stream);
    return a.is_u8 ? launch_i8<NTB, VGI_DOT, true, PRE, PIPE>(a, blocks, smem, stream) : launch_i8<NTB, VGI_DOT, false, PRE, PIPE>(a, blocks, smem, stream);
}
template <bool PRE>
static int launch_i8_ntb(const BatchArgsI8 &a, int ntb, int blocks, size_t smem, hipStream_t stream) {
    constexpr bool P = (VGI_PIPE != 0);
    if (ntb <= 8) return launch_i8_mode<8, PRE, P>(a, blocks, smem, stream);
#ifdef VGI_FEW                                                            // (measurement builds: rows of <= 256 and 513 .. 768 bytes, short lists)
    if (ntb > 16 && ntb <= 24 && vgi_pipelined(24, a.k)) return launch_i8_mode<24, PRE, P>(a, blocks, smem, stream);
    return -1;
#else
    if (ntb <= 16) return launch_i8_mode<16, PRE, P>(a, blocks, smem, stream);
    if (ntb <= 24) return vgi_pipelined(24, a.k) ? launch_i8_mode<24, PRE, P>(a, blocks, smem, stream) : launch_i8_mode<24, PRE, false>(a, blocks, smem, stream);
    if (ntb <= 32) return vgi_pipelined(32, a.k) ? launch_i8_mode<32, PRE, P>(a, blocks, smem, stream) : launch_i8_mode<32, PRE, false>(a, blocks, smem, stream);
    if (ntb <= 48) return launch_i8_mode<48, PRE, false>(a, blocks, smem, stream);
    return launch_i8_mode<64, PRE, false>(a, blocks, smem, stream);
#endif
}
#if defined(VGI_TU_PRE) || defined(VGI_TU_ALL)
extern "C" int vgi_launch_pre(const BatchArgsI8 *a, int ntb, int blocks, size_t smem, hipStream_t stream) {
    return launch_i8_ntb<true>(*a, ntb, blocks, smem, stream);
}
#endif
#ifndef VGI_TU_PRE
extern "C" int vgi_launch_real(const BatchArgsI8 *a, int ntb, int blocks, size_t smem, hipStream_t stream) {
    return launch_i8_ntb<false>(*a, ntb, blocks, smem, stream);
}

extern "C" int vg_batch_merge_launch(const uint64_t *dev_cand, int nq_pad, int lists_per_query, int npart, int k,
                                     uint64_t *dev_out_keys, hipStream_t stream);        // vg_batch.hip

static int vgi_ntb(long long stride_bytes) {                      // the instantiated k-step counts (32 bytes each)
    const int ntb = (int)((stride_bytes + 31) / 32);
    if (ntb <= 8) return 8;
    if (ntb <= 16) return 16;
    if (ntb <= 24) return 24;
    if (ntb <= 32) return 32;
    if (ntb <= 48) return 48;                                     // rows of 1 - 2 KiB: 4-wavefront workgroups
    if (ntb <= 64) return 64;
    return 0;
}

extern "C" int vg_batch_i8_queries_per_block(long long stride_bytes) { return VGI_WAVES_OF(vgi_ntb(stride_bytes)) * VGI_QPW; }

extern "C" size_t vg_batch_i8_lds_bytes(long long stride_bytes, int k) {
    const int NTB = vgi_ntb(stride_bytes);
    if (!NTB || k < 1 || k > VGI_MAX_K) return 0;
    const size_t b = vgi_lds(NTB, k, vgi_pipelined(NTB, k) ? VGI_PIPE_NBUF : 2);
    return b <= 160 * 1024 ? b : 0;
}

// dev_rows_signed: the corpus in signed representation; dev_queries: nq_pad x stride bytes (original representation).
// Returns 0, -1 if the shape is not served, a hipError_t otherwise.  dev_cand sized like the f32 kernel's.
extern "C" int vg_batch_i8_launch(const uint8_t *dev_rows_signed, long long n_rows, long long stride_bytes,
                                  const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root, int is_u8,
                                  const uint32_t *dev_row_stat, uint64_t *dev_cand, int npart,
                                  int tiles_per_part, uint64_t *dev_out_keys, hipStream_t stream) {
    const size_t smem = vg_batch_i8_lds_bytes(stride_bytes, k);
    if (!smem || nq_pad % vg_batch_i8_queries_per_block(stride_bytes) != 0 || npart < 1 || npart > VG_SEL_MAX_HEADS || n_rows < 1) return -1;
    if (mode < VGI_DOT || mode > VGI_L2 || !dev_row_stat) return -1;
    BatchArgsI8 a;
    a.rows = dev_rows_signed; a.queries = dev_queries; a.row_stat = dev_row_stat; a.cand = dev_cand;
    a.n_rows = n_rows; a.stride = stride_bytes; a.nq_pad = nq_pad; a.nq_real = nq_real; a.npart = npart; a.k = k;
    a.mode = mode; a.root = root; a.is_u8 = is_u8;
    const int ntb = (int)((stride_bytes + 31) / 32);
    const int G = nq_pad / vg_batch_i8_queries_per_block(stride_bytes);
    const int blocks = G * ((npart + 7) / 8) * 8;
    const long long ntiles = (n_rows + VGI_TILE - 1) / VGI_TILE;
    // Large corpora: the PRE pass over the first 1/512 of the rows (round 3; 1/32 before - one insert per query and tile) hands every query a
    // start threshold; the real pass scans every row from there.  Both write lists 0 .. npart-1 of the candidate buffer.
    long long pre = 0;
    {
        const int denom = vg_sw(SW_VG_BATCH_PREPASS, VGB_PREPASS_DENOM_DEFAULT);     // (vg_batch_common.h: re-measured in round 3)
        if (denom > 0 && ntiles >= 65536) pre = ((ntiles / denom + npart - 1) / npart) * npart;      // < 2M rows: one pass
    }
    int rc;
    a.npart_total = npart; a.part_base = 0; a.init_keys = nullptr; a.seed = 0;
    if (pre > 0) {
        a.tile_begin = 0; a.tile_end = pre; a.tiles_per_part = (int)(pre / npart);
        if ((rc = vgi_launch_pre(&a, ntb, blocks, smem, stream)) != 0) return rc;
        if ((rc = vg_batch_merge_launch(dev_cand, nq_pad, a.npart_total, npart, k, dev_out_keys, stream)) != 0) return rc;
        a.init_keys = dev_out_keys;
    }
    // the real pass, in stages over growing row ranges (vg_batch_common.h): the first one meets the pre-pass rows again (their
    // lists hold tile minima only), every later one starts from - and partition 0 carries on - the merged lists so far
    long long bounds[16];
    const int nstages = vgb_stage_bounds(ntiles, pre, bounds, 16);
    for (int s = 0; s < nstages; ++s) {
        a.tile_begin = bounds[s]; a.tile_end = bounds[s + 1];
        a.tiles_per_part = (int)((a.tile_end - a.tile_begin + npart - 1) / npart);
        a.seed = (s > 0) ? 1 : 0;
        if ((rc = vgi_launch_real(&a, ntb, blocks, smem, stream)) != 0) return rc;
        if ((rc = vg_batch_merge_launch(dev_cand, nq_pad, a.npart_total, npart, k, dev_out_keys, stream)) != 0) return rc;
        a.init_keys = dev_out_keys;
    }
    return 0;
}
#endif   // !VGI_TU_PRE
