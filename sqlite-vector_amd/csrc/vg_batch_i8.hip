// vg_batch_i8.hip - batched queries over a QUANTIZED corpus (uint8 / int8 rows, the vector_quantize format): Q x C^T
// on the integer matrix cores (v_mfma_i32_32x32x32_i8) with the same fused per-query top-k as the f32 kernel
// (vg_batch.hip).  Integer dot products are exact, so every distance is the single-query kernel's distance bit for
// bit (vg_accum.h AccumInt: exact 32-bit sums, one int->float conversion, correctly rounded sqrt / divide) and the
// result lists equal Q separate vector_quantize_scan calls exactly, ties included.
//
//   * a workgroup = 4 wavefronts owns 128 queries x one partition of the corpus; A (32 queries per wavefront) is
//     stationary in registers: lane (x, h) keeps bytes [32t + 16h, +16) of query x in a[t] (4 VGPRs per k-step);
//   * B streams through LDS in tiles of 32 rows by LDS-DMA (double buffered, transposed by 16-byte chunk - see the
//     kernel), one ds_read_b128 feeds one MFMA;
//   * uint8: the matrix core multiplies SIGNED bytes, so it runs on x' = x - 128 (a second, XOR-0x80 copy of the corpus
//     made once per corpus; queries are flipped while they are loaded) and the true dot product is restored exactly:
//         sum q x = sum q'x' + 128 (sum q + sum x) - 16384 L        (L = padded row length, pads are 0 <-> -128)
//     with sum x, sum x^2 per row from cached vectors (vg_i8_rowstat_kernel) and sum q, sum q^2 per query;
//   * gates are integer margins on the raw accumulator where the metric allows it (dot, L2: one v_add3 per register)
//     and a float margin for cosine (row norms differ per lane); survivors get the exact distance and go through the
//     same list insert as in vg_batch.hip; the two-pass launch (thresholds from a pre-pass) is the same as well.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "vg_batch_common.h"

typedef int vgi_i32x16 __attribute__((ext_vector_type(16)));
typedef int vgi_i32x4 __attribute__((ext_vector_type(4)));

#ifndef VGI_WAVES
#define VGI_WAVES 8                     // two wavefronts per SIMD: one gates / inserts / issues DMA while the other's MFMAs run
#endif
#define VGI_WAVES_LONG 4                // rows of 1 - 2 KiB: A alone is up to 256 registers, one wavefront per SIMD
#define VGI_WAVES_OF(NTB) ((NTB) <= 32 ? VGI_WAVES : VGI_WAVES_LONG)
#define VGI_QPW 32
#define VGI_TILE 32
#define VGI_MAX_K 32
#ifndef VGI_BPIPE
#define VGI_BPIPE 4                     // B-operand register quads in flight (LDS reads issued this many k-steps ahead)
#endif
#ifndef VGI_PHASED
#define VGI_PHASED 0                    // experiment, measured SLOWER (u8 cosine 15.0 vs 9.9 ms): the two wavefronts of a SIMD alternate, 3 tile buffers
#endif
#ifndef VGI_CHAINS
#define VGI_CHAINS 1                    // 2: even / odd k-steps accumulate in two independent chains (one wavefront can then issue an MFMA every 32 cycles)
#endif
#ifndef VGI_DEPTH2
#define VGI_DEPTH2 0                    // experiment, measured NEUTRAL (10.5 / 8.95 / 6.1 vs 9.9 / 9.3 / 5.5 ms): DMA two tiles ahead, counted vmcnt waits
#endif
#ifndef VGI_SKEW
#define VGI_SKEW 0                      // with VGI_ASYNC: the two wavefronts of a SIMD run half a tile apart
#endif
#ifndef VGI_ASYNC
#define VGI_ASYNC 0                     // experiment, measured NEUTRAL TO SLOWER (profiles/r2c_int8_batch_async_ring_vs_barrier.txt: u8 cosine 10.10 vs
                                        // 9.78 ms, dot 9.45 vs 9.24, D = 128 6.69 vs 5.87, D = 1024 12.80 vs 13.07; bit-exact tests pass): a tile ring with
                                        // per-buffer ready / free counters in LDS instead of one workgroup barrier per tile - the barrier is NOT what a tile
                                        // waits for
#endif
// tile buffers: the barrier schedule double-buffers; the async ring keeps 4 (3 for 1 KiB rows: LDS) and lets a wavefront run
// up to two (one) tiles ahead of the slowest one; rows beyond 1 KiB (4-wavefront workgroups, 48 / 64 KiB tiles) keep the barrier
#define VGI_IS_ASYNC(NTB) (VGI_ASYNC && (NTB) <= 32)
#define VGI_NBUF_OF(NTB) (VGI_IS_ASYNC(NTB) ? ((NTB) <= 24 ? 4 : 3) : ((VGI_PHASED || VGI_DEPTH2) ? 3 : 2))

enum { VGI_DOT = 0, VGI_COS = 1, VGI_L2 = 2 };

#ifndef VGI_TIMING
#define VGI_TIMING 0                    // measurement builds (tools/tools_i8_timing.py, tools/build_i8_variants.sh timing -DVGI_TIMING=1)
#endif
#if VGI_TIMING
// s_memtime ticks summed over all wavefronts of the REAL pass (barrier schedule): k loop (incl. the MFMA drain) | gate math |
// inserts | DMA wait | barrier | whole loop | wave-tiles | wave-tiles with pending registers
__device__ unsigned long long vgi_ticks[8];
extern "C" int vg_batch_i8_timing(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(vgi_ticks), sizeof(vgi_ticks)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vgi_ticks), z, sizeof(z)) != hipSuccess) return -1; }
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
    const int32_t *row_sx;    // sum x per row (original representation)
    const uint32_t *row_sxx;  // sum x^2 per row
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

// NTB = 32-byte k-steps per row (rows up to NTB * 32 bytes)
// PRE = the pre-pass variant (large corpora): only the SMALLEST distance of each query in a tile enters its list.  k
// entries then stand for k different rows, so the k-th of them bounds the query's final k-th best distance from above:
// the start threshold of the real pass, which scans EVERY row.  A list warms up with one insert per (query, tile)
// instead of one per passing row (each insert is an LDS round trip of the whole wavefront).
template <int NTB, int MODE, bool IS_U8, bool PRE>
__global__ __launch_bounds__(64 * VGI_WAVES_OF(NTB), 1) void vg_batch_i8_kernel(BatchArgsI8 a) {
    constexpr int WAVES = VGI_WAVES_OF(NTB), THREADS = 64 * WAVES, QPB = WAVES * VGI_QPW;
        constexpr bool COS = (MODE == VGI_COS), L2M = (MODE == VGI_L2);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // LDS tile, TRANSPOSED BY 16-BYTE CHUNK: chunk column c of the 32 rows is one contiguous 512-byte run
    // (address c * 512 + row * 16).  A lane's b128 read of (row x, chunk 2t + h) sits next to its neighbours' - no
    // bank conflicts, no row padding - and one LDS-DMA instruction (64 lanes x 16 bytes = two chunk columns) carries
    // 64 (row, chunk) pairs whatever the row length.  With one instruction per ROW (the f32 kernel's layout) a
    // 128-byte row used 8 of the 64 lanes, and the ~100-cycle issue cost of an LDS-DMA instruction made the kernel's
    // time proportional to the row COUNT, not to the bytes.
    constexpr int TILE_BYTES = NTB * 2 * 512;
    constexpr int L = NTB * 32;                                 // padded row length the matrix core sees
    constexpr int NBUF = VGI_NBUF_OF(NTB);
    constexpr bool ASYNC = VGI_IS_ASYNC(NTB);
    uint8_t *tile0 = smem;
    uint32_t *rstat_lds = reinterpret_cast<uint32_t *>(smem + NBUF * TILE_BYTES);        // [buffers][sum x: 32 | sum x^2: 32]
    uint32_t *qstat_lds = rstat_lds + NBUF * 64;                                         // [waves][32][2]: sum q, sum q^2
    uint64_t *lists = reinterpret_cast<uint64_t *>(qstat_lds + WAVES * VGI_QPW * 2);  // [4][32][k]
    uint32_t *ring_ctr = reinterpret_cast<uint32_t *>(lists + (size_t)WAVES * VGI_QPW * a.k);   // async ring: ready[NBUF] | freed[NBUF]

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
    uint64_t *wave_lists = lists + (size_t)wave * VGI_QPW * k;
    {
        const uint32_t sq = sq_part + __shfl_xor(sq_part, 32), sqq = sqq_part + __shfl_xor(sqq_part, 32);
        if (h == 0) { qs_w[2 * x] = sq; qs_w[2 * x + 1] = sqq; }
        const bool seeded = a.seed != 0 && part == 0;               // (keys of rows no later stage meets again)
        for (int s = lane; s < VGI_QPW * k; s += 64)
            wave_lists[s] = seeded ? a.init_keys[(long long)(q0 + s / k) * 64 + s % k] : VG_EMPTY_KEY;
    }
    // pad columns never touched by the DMA must read as "0" of the original representation
    for (int s = tid; s < NBUF * TILE_BYTES / 4; s += THREADS) reinterpret_cast<uint32_t *>(tile0)[s] = IS_U8 ? 0x80808080u : 0u;
    if (ASYNC && tid < 2 * NBUF) ring_ctr[tid] = (tid == 0) ? (uint32_t)WAVES : 0u;     // tile 0 is "ready" after the prologue's barrier
    __syncthreads();

    // ---- tile streaming by LDS-DMA: piece p = chunk columns 2p and 2p+1 of all 32 rows; wavefront w moves pieces
    // w, w + WAVES, ...  Lane l reads 16 bytes of row (l & 31) at chunk 2p + (l >> 5); it lands at M0 + 16 * l.
    const int chunks_per_row = (int)(a.stride / 16);
    const int npieces = (chunks_per_row + 1) / 2;               // pieces that carry data (<= NTB)
    const long long tile_first = a.tile_begin + (long long)part * a.tiles_per_part;
    const long long tile_last = min(tile_first + a.tiles_per_part, a.tile_end);
    const unsigned long long stride_b = (unsigned long long)a.stride;
    const uint32_t lds_tile0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)tile0;
    constexpr int NPIECE = (NTB + WAVES - 1) / WAVES;  // piece slots per wavefront and tile (compile time)
    uint64_t piece_mask[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int p = wave + i * WAVES;
#ifdef VGI_ABLATE_HALF_DMA      // measurement build (WRONG results): only every second piece of a tile is fetched
        piece_mask[i] = __ballot(p < npieces && (p & 1) == 0 && (2 * p + h) < chunks_per_row);
#else
        piece_mask[i] = __ballot(p < npieces && (2 * p + h) < chunks_per_row);
#endif
    }
    // rows past the end of the corpus (last tile only) re-read the last row: their scores are masked by the row bound
    // (the tile-major copy holds whole tiles: rows past the end of the corpus read whatever the allocation holds there, their
    //  scores are masked by the row bound)
    auto lane_offset = [&](long long) -> uint32_t { return (uint32_t)lane * 16u; };
    // The per-row sums of a tile ride the same pipeline (two 128-byte pieces, issued by the last wavefront): read
    // with ordinary loads at the tile boundary they cost a full L2 / HBM round trip per tile - with only 8..32
    // MFMAs per tile that latency WAS the kernel time (7 ms of the 10.7 at D = 768, and independent of D).
    const uint32_t lds_rstat0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)rstat_lds;
    const uint64_t stat_mask = __ballot(wave == WAVES - 1 && lane < 8);
    const uint32_t stat_goff = (uint32_t)lane * 16u;
    auto dma_stats = [&](long long tile, int buf) {
        const uint8_t *b0 = reinterpret_cast<const uint8_t *>(a.row_sx + tile * VGI_TILE);
        const uint8_t *b1 = reinterpret_cast<const uint8_t *>(a.row_sxx + tile * VGI_TILE);
        const uint32_t d0 = lds_rstat0 + (uint32_t)(buf * 256), d1 = d0 + 128u;
        uint32_t keep;
        uint64_t keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %7\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keep_exec) : "v"(stat_goff), "s"(b0), "s"(b1), "s"(d0), "s"(d1), "s"(stat_mask) : "memory", "scc");
    };
    auto dma_piece = [&](long long tile, uint32_t lane_goff, int buf, int i) {
        const int p = wave + i * WAVES;
        const uint8_t *sbase = a.rows + (unsigned long long)(tile * VGI_TILE) * stride_b + (unsigned)p * 1024u;   // 1 KiB contiguous
        const uint32_t lds_dst = lds_tile0 + (uint32_t)(buf * TILE_BYTES + p * 1024);
        uint32_t keep;
        uint64_t keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_and_b64 exec, exec, %5\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keep_exec) : "v"(lane_goff), "s"(sbase), "s"(lds_dst), "s"(piece_mask[i]) : "memory", "scc");
    };

    // ---- per-register query state (register r of lane (x, h) belongs to query qi(r, h) = (r&3) + 8*(r>>2) + 4*h)
    //   qq_reg   sum q^2                          cq_reg   what turns the raw accumulator into sum q x, query part
    //   thr_reg  current k-th best distance       gate_i   integer gate (dot, L2)     gate_f  float gate (cosine)
    uint32_t qq_reg[16];
    int cq_reg[16], gate_i[16];
    float thr_reg[16], gate_f[16], na_reg[16];
#ifndef VGI_FOLD
#define VGI_FOLD 1                      // 0: the round-1 boundary for every metric (accumulators start at 0, 3-5 VALU operations per register and tile)
#endif
    // (measured, profiles/r2g: dot 9.05 -> 9.01 ms - its tile then waited longer for the DMA instead, until the tile-major copy -, L2 9.44 -> 9.10,
    //  D = 1024 13.6 -> 12.2; a first cosine variant - the lane's LOOSEST gate as a pre-test - was slower, 9.78 -> 10.11: it fired too often)
    constexpr bool FOLD = (VGI_FOLD != 0);
    // FOLDED GATES: the accumulator of register r STARTS at acc_init[r] instead of 0 (integer sums: exact), so that the tile
    // boundary needs ONE max over the 16 registers and one comparison per lane instead of 3-5 operations per register:
    //   dot     init = -gate_i                  any pair passes  <=>  max_r acc' + cx >= 0
    //   L2      init = -(gate_i >> 1)           any pair passes   =>  max_r acc' - ((xx - 2 cx) >> 1) >= 0     (floors: a superset)
    //   cosine  init = cq + 1 (query part of q.x + 1): q.x >= G nb - 1  <=>  (q.x + 1) / G >= nb for a closed gate G > 0, so
    //           any pair passes  <=>  max_r float(acc' + cx) * (1 / G_r) >= nb   - 3 operations per register instead of 5, the 16 results
    //           are comparable (one max tree, one comparison with the lane's row norm); registers whose gate is still open (G <= 0:
    //           list not full) are flagged in open_mask and always pass
    // The exact distance test in reg_insert is unchanged (it gets the raw accumulator acc' - init back).
    int acc_init[16];
    float invg[16];                                      // cosine: (1 + 1e-6) / gate_f of a closed gate
    uint32_t open_mask = 0;                              // cosine: registers whose gate is open (accept everything)
    const bool l2_root = a.root != 0;
    auto as_float_like = [](uint32_t v) -> float { return IS_U8 ? (float)v : (float)(int32_t)v; };
    // gates are supersets of "distance <= thr" (exact test in reg_insert):
    //   dot     -(float)qx <= thr          <=  qx >= floor(-thr) - 8             margin = acc + cx - (G - cq)
    //   L2      (float)(qq+xx-2qx) <= thr2 <=  qq+xx-2qx <= ceil(thr2*(1+1e-6)) + 8   margin = 2 acc - (qq - 2cq - T) - (xx - 2cx)
    //   cosine  1 - qx/(na nb) <= thr      <=  (float)qx >= (1-thr) na nb (1 -+ 1e-5) - 1
    auto set_gate = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const float thr = thr_reg[r];
        if (COS) {
            const float Gf = (1.0f - thr) * na_reg[r];
            gate_f[r] = fmaxf(Gf - 1e-5f * fabsf(Gf), -3.0e38f);
        } else if (L2M) {
            const float thr2 = l2_root ? thr * thr : thr;
            float Tf = thr2 * (1.0f + 1e-6f) + 8.0f;
            const int T = (Tf < 1.5e9f) ? (int)Tf : 1500000000;           // +Inf / NaN: accept everything (totals < 2^27)
            gate_i[r] = (int)qq_reg[r] - 2 * cq_reg[r] - T;
        } else {
            float Gf = -thr - 8.0f;
            const int Gq = (Gf > -1.5e9f) ? (int)floorf(Gf) : -1500000000;  // thr = +Inf: accept everything (|qx| < 2^27)
            gate_i[r] = Gq - cq_reg[r];
        }
        acc_init[r] = !FOLD ? 0 : (COS ? cq_reg[r] + 1 : (L2M ? -(gate_i[r] >> 1) : -gate_i[r]));
        if (COS) {
            const bool closed = gate_f[r] > 1e-30f;
            invg[r] = closed ? __fdividef(1.0f + 1e-6f, gate_f[r]) : 0.0f;
            open_mask = closed ? (open_mask & ~(1u << r)) : (open_mask | (1u << r));
        }
    };

    vgb_static_for<0, 16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const uint32_t sq = qs_w[2 * qi], sqq = qs_w[2 * qi + 1];
        qq_reg[r] = sqq;
        cq_reg[r] = IS_U8 ? (int)(128u * sq) - 16384 * L : 0;
        na_reg[r] = sqrtf(as_float_like(sqq));
        // the pre-pass bound is the distance of an actual row, which this pass meets again: one ulp up, so that the
        // strict comparison below lets a row AT the bound in
        thr_reg[r] = a.init_keys ? nextafterf(vgb_kth_distance(a.init_keys[(long long)(q0 + qi) * 64 + (k - 1)]), INFINITY) : INFINITY;
        set_gate(rc);
        if (q0 + qi >= a.nq_real) {                      // padding (an all-zero query ties every row at cosine 1.0)
            thr_reg[r] = -INFINITY;
            if (COS) gate_f[r] = 3.0e38f; else gate_i[r] = 1500000000;
            acc_init[r] = !FOLD ? 0 : (COS ? cq_reg[r] + 1 : (L2M ? -(gate_i[r] >> 1) : -gate_i[r]));
            if (COS) { invg[r] = 0.0f; open_mask &= ~(1u << r); }
        }
    });

    // exact distance of one (query, row) pair from the raw accumulator - the single-query kernel's epilogue (vg_accum.h)
    auto reg_distance = [&](auto rc, int acc_r, int cx, uint32_t xx) -> float {
        constexpr int r = decltype(rc)::value;
        const uint32_t qx = (uint32_t)(acc_r + cq_reg[r] + cx);            // sum q x, modulo 2^32 like the reference
        float d;
        if (COS) {
            d = vg_cosine_from_norms(as_float_like(qx), na_reg[r], sqrtf(as_float_like(xx)));
        } else if (L2M) {
            const float t = (float)(uint32_t)(qq_reg[r] + xx - 2u * qx);
            d = l2_root ? sqrtf(t) : t;
        } else {
            d = -as_float_like(qx);
        }
        return vg_clamp(d);
    };
    auto reg_insert = [&](auto rc, int acc_r, long long row, int cx, uint32_t xx) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int q_lo = (r & 3) + 8 * (r >> 2);
        const float d = reg_distance(rc, acc_r, cx, xx);
        // strict: rows arrive in scan order, a row that only ties the k-th best has the larger position and loses
        const bool pass = (row < a.n_rows) && (d < thr_reg[r]);
        unsigned long long m = __ballot(pass);
        if constexpr (PRE) {
            if (m) {
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
                    if (h == hh) {
                        thr_reg[r] = fminf(nt, thr_reg[r]);
                        set_gate(rc);
                    }
                }
            }
            return;
        }
        const uint64_t key = vg_make_key(d, (uint32_t)row);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int hh = src >> 5;
            uint64_t *list = wave_lists + (q_lo + 4 * hh) * k;
            const uint64_t c = vg_readlane64(key, src);
            const float nt = vgb_kth_distance(vgb_list_insert(list, k, lane, c));
            if (h == hh) {
                thr_reg[r] = fminf(nt, thr_reg[r]);
                set_gate(rc);
            }
        }
    };

    constexpr int BP = VGI_BPIPE < NTB ? VGI_BPIPE : NTB;
    vgi_i32x4 bq[BP];
    vgi_i32x16 acc;
    // the k loop of one tile: nothing but MFMAs, the B-operand LDS reads and the DMA issue of a later tile
    auto k_loop = [&](int cur_buf, long long tile_next, int next_buf) __attribute__((always_inline)) {
        const uint32_t goff_next = lane_offset(tile_next);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc_init[r];
#if VGI_CHAINS == 2
        vgi_i32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0;
#endif
        const uint32_t baddr = lds_tile0 + (uint32_t)(cur_buf * TILE_BYTES + h * 512 + x * 16);
        vgb_static_for<0, BP>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            vgi_lds_read128<1024 * t>(bq[t], baddr);
        });
        vgb_static_for<0, NTB>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int in_flight_after = (NTB - 1 - t) < (BP - 1) ? (NTB - 1 - t) : (BP - 1);
            vgi_wait_lds<in_flight_after>(bq[t % BP]);
            const vgi_i32x4 b = bq[t % BP];
#if VGI_CHAINS == 2
            if constexpr (t & 1) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[t], b, acc2, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[t], b, acc, 0, 0, 0);
#else
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[t], b, acc, 0, 0, 0);
#endif
            if constexpr (t + BP < NTB) vgi_lds_read128<1024 * (t + BP)>(bq[t % BP], baddr);
            // the DMA pieces of the later tile, spread over the first half of the k loop
            constexpr int NTD = (NTB + 1) / 2;
            constexpr int pc_lo = (t >= NTD) ? NPIECE : (t * NPIECE + NTD - 1) / NTD;
            constexpr int pc_hi = (t >= NTD) ? NPIECE : (t + 1 == NTD ? NPIECE : ((t + 1) * NPIECE + NTD - 1) / NTD);
            vgb_static_for<pc_lo, pc_hi>([&](auto pcc) { dma_piece(tile_next, goff_next, next_buf, decltype(pcc)::value); });
            if constexpr (t == 0) dma_stats(tile_next, next_buf);
        });
#if VGI_CHAINS == 2
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];                   // (exact: integer sums)
#endif
    };
    // the tile boundary: margins (integer for dot / L2, float for cosine), one ballot, the rare inserts
#if VGI_TIMING
    unsigned long long tk_gate_end = 0, tk_pend = 0, tk_k = 0, tk_gate = 0, tk_ins = 0, tk_dma = 0, tk_bar = 0, tk_tiles = 0;
#endif
    auto boundary_with = [&](long long tile, int sx, uint32_t xx) __attribute__((always_inline)) {
        const long long row_cur = tile * VGI_TILE + x;
        const int cx = IS_U8 ? 128 * sx : 0;
        unsigned pend = 0;
        bool any;
        if constexpr (FOLD && COS) {
            const float nb = sqrtf(as_float_like(xx));
            float fmax = -INFINITY;
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                fmax = fmaxf(fmax, as_float_like((uint32_t)(acc[r] + cx)) * invg[r]);       // (q.x + 1) / G_r
            });
            any = __ballot(open_mask != 0u || fmax >= nb) != 0;
            if (any) {
                vgb_static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const float qxf = as_float_like((uint32_t)(acc[r] - 1 + cx));
                    pend |= __ballot(qxf + 1.0f - gate_f[r] * nb >= 0.0f) ? (1u << r) : 0u;
                });
            }
        } else if constexpr (FOLD) {
            int amax = acc[0];                           // max over the lane's 16 folded accumulators: 8 x v_max3
            vgb_static_for<1, 16>([&](auto rc) { constexpr int r = decltype(rc)::value; amax = acc[r] > amax ? acc[r] : amax; });
            const int h2 = L2M ? (((int)xx - 2 * cx) >> 1) : -cx;
            any = __ballot(amax - h2 >= 0) != 0;
            if (any) {
                vgb_static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    pend |= __ballot(acc[r] - h2 >= 0) ? (1u << r) : 0u;
                });
            }
        } else
        if (COS) {
            const float nb = sqrtf(as_float_like(xx));
            float margin = -INFINITY;
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const float qxf = as_float_like((uint32_t)(acc[r] + cq_reg[r] + cx));
                margin = fmaxf(margin, qxf + 1.0f - gate_f[r] * nb);
            });
            any = __ballot(margin >= 0.0f) != 0;
            if (any) {
                vgb_static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const float qxf = as_float_like((uint32_t)(acc[r] + cq_reg[r] + cx));
                    pend |= __ballot(qxf + 1.0f - gate_f[r] * nb >= 0.0f) ? (1u << r) : 0u;
                });
            }
        } else {
            const int hx = L2M ? (int)xx - 2 * cx : -cx;
            int margin = -0x7FFFFFFF;
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int m = (L2M ? 2 * acc[r] : acc[r]) - gate_i[r] - hx;
                margin = m > margin ? m : margin;
            });
            any = __ballot(margin >= 0) != 0;
            if (any) {
                vgb_static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    pend |= __ballot((L2M ? 2 * acc[r] : acc[r]) - gate_i[r] - hx >= 0) ? (1u << r) : 0u;
                });
            }
        }
#if VGI_TIMING
        tk_gate_end = __builtin_readcyclecounter();
        tk_pend += pend ? 1 : 0;
#endif
        if (pend) {
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if (pend & (1u << r)) reg_insert(rc, acc[r] - acc_init[r], row_cur, cx, xx);     // (the raw accumulator)
            });
        }
    };

    auto boundary = [&](long long tile, int cur_buf) __attribute__((always_inline)) {
        // this tile's row sums (landed with the tile)
        boundary_with(tile, (int)rstat_lds[cur_buf * 64 + x], rstat_lds[cur_buf * 64 + 32 + x]);
    };

    if constexpr (ASYNC) {
        // ASYNC RING.  Tile t lives in buffer t % NBUF; the k loop of tile t issues this wavefront's DMA pieces of tile t + 2.
        // No workgroup barrier in the loop: ready[b] counts the wavefronts whose pieces of the tile in buffer b have landed,
        // freed[b] those that are done reading it (both only ever grow: generation g of a buffer is complete at WAVES * (g + 1)).
        // A wavefront in its survivor path no longer stops the other seven: they run on until they need a buffer it still
        // reads - NBUF - 2 tiles later.
        volatile uint32_t *ready = ring_ctr, *freed = ring_ctr + NBUF;
        auto wait_ge = [&](volatile uint32_t *p, uint32_t target) __attribute__((always_inline)) {
            while ((uint32_t)__builtin_amdgcn_readfirstlane((int)*p) < target) __builtin_amdgcn_s_sleep(1);
        };
        int n_mine = (stat_mask != 0) ? 2 : 0;                           // DMA instructions this wavefront issues per tile
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) n_mine += (piece_mask[i] != 0) ? 1 : 0;
        if (tile_first < tile_last) {
            const long long t1 = min(tile_first + 1, tile_last - 1);
            const uint32_t goff0 = lane_offset(tile_first), goff1 = lane_offset(t1);
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) dma_piece(tile_first, goff0, 0, pc);
            dma_stats(tile_first, 0);
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) dma_piece(t1, goff1, 1, pc);
            dma_stats(t1, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                 // tiles 0 and 1 have landed; ready[0] was preset, ready[1] is signalled in iteration 0
#if VGI_SKEW
        // Start the second wavefront of every SIMD (waves WAVES/2 ..: the same SIMDs as waves 0 .. WAVES/2-1) half a tile late: nothing
        // in the ring re-aligns them afterwards, so while one of a SIMD's two wavefronts is at its boundary the other is in its k loop.
        if (wave >= WAVES / 2) wait_ge(&freed[0], (uint32_t)(WAVES / 2));
#endif
        for (long long tile = tile_first; tile < tile_last; ++tile) {
            const long long ti = tile - tile_first;
            const int b = (int)(ti % NBUF), nb = (int)((ti + 2) % NBUF);
            const uint32_t g = (uint32_t)(ti / NBUF), g2 = (uint32_t)((ti + 2) / NBUF);
            wait_ge(&ready[b], (uint32_t)WAVES * (g + 1u));              // every wavefront's pieces of this tile are in LDS
            wait_ge(&freed[nb], (uint32_t)WAVES * g2);                   // nobody still reads the buffer tile t + 2 goes to
            k_loop(b, min(tile + 2, tile_last - 1), nb);
            const int sx = (int)rstat_lds[b * 64 + x];
            const uint32_t xx = rstat_lds[b * 64 + 32 + x];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the two reads above have returned)
            if (lane == 0) atomicAdd((uint32_t *)&freed[b], 1u);         // done with buffer b - BEFORE the boundary's slow path
            // my pieces of tile t + 1 (issued one iteration ago) have landed: everything but this k loop's own DMA issue.
            // Signalled BEFORE the boundary as well, so that nobody waits for this wavefront's inserts.
            switch (n_mine) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            }
            if (lane == 0) atomicAdd((uint32_t *)&ready[(ti + 1) % NBUF], 1u);
            boundary_with(tile, sx, xx);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#if VGI_PHASED
    // PHASED schedule (experiment, slower).  The workgroup's wavefronts form two groups (waves 0-3 / 4-7: one of each per SIMD).  A step is
    // one barrier interval; on even steps group 0 runs the k loop of tile s/2 while group 1 is at the boundary of the
    // tile it multiplied one step earlier, on odd steps the roles swap - the matrix pipe of a SIMD always has a
    // wavefront in its k loop instead of both idling at the boundary together.  Tile t sits in buffer t % 3; during its
    // k loop of tile t group 0 fetches its DMA pieces of tile t+1, group 1 (one step later) its pieces of tile t+2;
    // everybody waits for its own DMA at the end of its BOUNDARY step, a full step after issuing it.
    {
        const int grp = wave >> 2;
        const long long T = tile_last - tile_first;
        if (T > 0) {
            const uint32_t goff0 = lane_offset(tile_first);
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) dma_piece(tile_first, goff0, 0, pc);
            dma_stats(tile_first, 0);
            if (grp == 1) {
                const long long t1 = min(tile_first + 1, tile_last - 1);
                const uint32_t goff1 = lane_offset(t1);
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc) dma_piece(t1, goff1, 1, pc);
                dma_stats(t1, 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (long long s = 0; s <= 2 * T; ++s) {
            const bool kl = (grp == 0) ? ((s & 1) == 0 && s < 2 * T) : ((s & 1) == 1);
            const bool bd = (grp == 0) ? ((s & 1) == 1) : ((s & 1) == 0 && s >= 2);
            if (kl) {
                const long long ti = (s - grp) >> 1;
                k_loop((int)(ti % 3), min(tile_first + ti + 1 + grp, tile_last - 1), (int)((ti + 1 + grp) % 3));
            } else if (bd) {
                const long long ti = (s - 1 - grp) >> 1;
                boundary(tile_first + ti, (int)(ti % 3));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }
#elif VGI_DEPTH2
    // Prefetch distance TWO tiles (experiment: the hypothesis "a tile costs the DMA's latency plus its transfer time" was
    // wrong - no gain, so the DMA is not what a tile waits for).  The k loop of tile t issues the DMA of tile t+2 (three
    // buffers) and the wait at the end of tile t leaves exactly those instructions outstanding (loads return in order).
    {
        int n_mine = (stat_mask != 0) ? 2 : 0;                           // DMA instructions this wavefront issues per tile
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) n_mine += (piece_mask[i] != 0) ? 1 : 0;
        if (tile_first < tile_last) {
            const long long t1 = min(tile_first + 1, tile_last - 1);
            const uint32_t goff0 = lane_offset(tile_first), goff1 = lane_offset(t1);
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) dma_piece(tile_first, goff0, 0, pc);
            dma_stats(tile_first, 0);
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) dma_piece(t1, goff1, 1, pc);
            dma_stats(t1, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (long long tile = tile_first; tile < tile_last; ++tile) {
            const long long ti = tile - tile_first;
            const int cur_buf = (int)(ti % 3);
            k_loop(cur_buf, min(tile + 2, tile_last - 1), (int)((ti + 2) % 3));
            boundary(tile, cur_buf);
            switch (n_mine) {                                            // everything but this tile's own DMA issue has landed
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            }
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#else
    if (tile_first < tile_last) {
        const uint32_t goff0 = lane_offset(tile_first);
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) dma_piece(tile_first, goff0, 0, pc);
        dma_stats(tile_first, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    VGI_TICK(t_loop0);
    for (long long tile = tile_first; tile < tile_last; ++tile) {
        const int cur_buf = (int)((tile - tile_first) & 1);
        VGI_TICK(t0);
        k_loop(cur_buf, min(tile + 1, tile_last - 1), cur_buf ^ 1);      // (the last iteration re-fetches its own tile)
#if VGI_TIMING
        asm volatile("s_nop 0" :: "v"(acc[15]));                          // the k loop's last MFMA has retired
#endif
        VGI_TICK(t1);
        boundary(tile, cur_buf);
        VGI_TICK(t3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VGI_TICK(t4);
        __syncthreads();
#if VGI_TIMING
        const unsigned long long t5 = __builtin_readcyclecounter();
        tk_k += t1 - t0; tk_gate += tk_gate_end - t1; tk_ins += t3 - tk_gate_end; tk_dma += t4 - t3; tk_bar += t5 - t4; ++tk_tiles;
#endif
    }
#if VGI_TIMING
    if (!PRE && lane == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        atomicAdd(&vgi_ticks[0], tk_k); atomicAdd(&vgi_ticks[1], tk_gate); atomicAdd(&vgi_ticks[2], tk_ins); atomicAdd(&vgi_ticks[3], tk_dma);
        atomicAdd(&vgi_ticks[4], tk_bar); atomicAdd(&vgi_ticks[5], t_end - t_loop0); atomicAdd(&vgi_ticks[6], tk_tiles); atomicAdd(&vgi_ticks[7], tk_pend);
    }
#endif
#endif
    }   // !ASYNC

    for (int s = lane; s < VGI_QPW * 64; s += 64) {
        const int qi = s >> 6, slot = s & 63;
        a.cand[((long long)(q0 + qi) * a.npart_total + a.part_base + part) * 64 + slot] = (slot < k) ? wave_lists[qi * k + slot] : VG_EMPTY_KEY;
    }
}

#ifndef VGI_TU_PRE
// ---- per-row sums of the ORIGINAL representation + the tile-major copy the matrix core reads (uint8: XOR 0x80)
template <bool IS_U8>
__global__ __launch_bounds__(256) void vg_i8_rowstat_kernel(const uint8_t *rows, long long row0, long long n, long long stride,
                                                            int32_t *sx, uint32_t *sxx, uint8_t *tiled) {
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
        if (sub == 0) { sx[row0 + r] = (int32_t)s1; sxx[row0 + r] = s2; }
    }
}

extern "C" int vg_i8_rowstat_launch(const uint8_t *dev_rows, long long row0, long long n, long long stride, int is_u8,
                                    int32_t *dev_sx, uint32_t *dev_sxx, uint8_t *dev_flipped, hipStream_t stream) {
    if (n <= 0) return 0;
    long long blocks = (n * 16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (is_u8) hipLaunchKernelGGL((vg_i8_rowstat_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride, dev_sx, dev_sxx, dev_flipped);
    else hipLaunchKernelGGL((vg_i8_rowstat_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride, dev_sx, dev_sxx, dev_flipped);
    return (int)hipGetLastError();
}

#endif   // !VGI_TU_PRE

// ---- host side
// The kernel instantiations are split over two translation units compiled from this file (build.py): the real-pass
// kernels here, the PRE kernels with -DVGI_TU_PRE (vg_batch_i8_pre.o); -DVGI_TU_ALL: both in this one unit (the
// measurement builds of tools/build_i8_variants.sh).
extern "C" int vgi_launch_pre(const BatchArgsI8 *a, int ntb, int blocks, size_t smem, hipStream_t stream);
extern "C" int vgi_launch_real(const BatchArgsI8 *a, int ntb, int blocks, size_t smem, hipStream_t stream);

template <int NTB, int MODE, bool IS_U8, bool PRE>
static int launch_i8(const BatchArgsI8 &a, int blocks, size_t smem, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vg_batch_i8_kernel<NTB, MODE, IS_U8, PRE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((vg_batch_i8_kernel<NTB, MODE, IS_U8, PRE>), dim3((unsigned)blocks), dim3(64 * VGI_WAVES_OF(NTB)), smem, stream, a);
    return (int)hipGetLastError();
}
template <int NTB, int MODE, bool PRE>
static int launch_i8_sign(const BatchArgsI8 &a, int blocks, size_t smem, hipStream_t stream) {
    return a.is_u8 ? launch_i8<NTB, MODE, true, PRE>(a, blocks, smem, stream) : launch_i8<NTB, MODE, false, PRE>(a, blocks, smem, stream);
}
template <int NTB, bool PRE>
static int launch_i8_mode(const BatchArgsI8 &a, int blocks, size_t smem, hipStream_t stream) {
    if (a.mode == VGI_COS) return launch_i8_sign<NTB, VGI_COS, PRE>(a, blocks, smem, stream);
    if (a.mode == VGI_L2) return launch_i8_sign<NTB, VGI_L2, PRE>(a, blocks, smem, stream);
    return launch_i8_sign<NTB, VGI_DOT, PRE>(a, blocks, smem, stream);
}
template <bool PRE>
static int launch_i8_ntb(const BatchArgsI8 &a, int ntb, int blocks, size_t smem, hipStream_t stream) {
    if (ntb <= 8) return launch_i8_mode<8, PRE>(a, blocks, smem, stream);
    if (ntb <= 16) return launch_i8_mode<16, PRE>(a, blocks, smem, stream);
    if (ntb <= 24) return launch_i8_mode<24, PRE>(a, blocks, smem, stream);
    if (ntb <= 32) return launch_i8_mode<32, PRE>(a, blocks, smem, stream);
#if !VGI_PHASED                                                       // (the phased experiment assumes 8 wavefronts)
    if (ntb <= 48) return launch_i8_mode<48, PRE>(a, blocks, smem, stream);
    return launch_i8_mode<64, PRE>(a, blocks, smem, stream);
#else
    return -1;
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
    if (VGI_PHASED && NTB > 32) return 0;
    const size_t waves = (size_t)VGI_WAVES_OF(NTB);
    const size_t b = (size_t)VGI_NBUF_OF(NTB) * (NTB * 1024 + 256) + waves * VGI_QPW * 2 * 4 + waves * VGI_QPW * k * 8 + 64;   // (+ the ring counters)
    return b <= 160 * 1024 ? b : 0;
}

// dev_rows_signed: the corpus in signed representation; dev_queries: nq_pad x stride bytes (original representation).
// Returns 0, -1 if the shape is not served, a hipError_t otherwise.  dev_cand sized like the f32 kernel's.
extern "C" int vg_batch_i8_launch(const uint8_t *dev_rows_signed, long long n_rows, long long stride_bytes,
                                  const uint8_t *dev_queries, int nq_pad, int nq_real, int k, int mode, int root, int is_u8,
                                  const int32_t *dev_sx, const uint32_t *dev_sxx, uint64_t *dev_cand, int npart,
                                  int tiles_per_part, uint64_t *dev_out_keys, hipStream_t stream) {
    const size_t smem = vg_batch_i8_lds_bytes(stride_bytes, k);
    if (!smem || nq_pad % vg_batch_i8_queries_per_block(stride_bytes) != 0 || npart < 1 || npart > VG_SEL_MAX_HEADS || n_rows < 1) return -1;
    if (mode < VGI_DOT || mode > VGI_L2 || !dev_sx || !dev_sxx) return -1;
    BatchArgsI8 a;
    a.rows = dev_rows_signed; a.queries = dev_queries; a.row_sx = dev_sx; a.row_sxx = dev_sxx; a.cand = dev_cand;
    a.n_rows = n_rows; a.stride = stride_bytes; a.nq_pad = nq_pad; a.nq_real = nq_real; a.npart = npart; a.k = k;
    a.mode = mode; a.root = root; a.is_u8 = is_u8;
    const int ntb = (int)((stride_bytes + 31) / 32);
    const int G = nq_pad / vg_batch_i8_queries_per_block(stride_bytes);
    const int blocks = G * ((npart + 7) / 8) * 8;
    const long long ntiles = (n_rows + VGI_TILE - 1) / VGI_TILE;
    // Large corpora: the PRE pass over the first 1/32 of the rows (one insert per query and tile) hands every query a
    // start threshold; the real pass scans every row from there.  Both write lists 0 .. npart-1 of the candidate buffer.
    long long pre = 0;
    {
        const char *e = getenv("VG_BATCH_PREPASS");
        const int denom = (e && *e) ? atoi(e) : 32;
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
