// vg_batch.hip - batched queries: Q x Corpus^T as a dense f32 GEMM on the matrix cores with a fused per-query
// top-k (BASELINE config C5: 1024 queries x 10M x 384 f32, dot, top-20).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate = a k-ordered fmaf chain, exact f32 semantics, 157 TF
// peak).  The 1024 x 10M score matrix (41 GB) is never materialised: each 32x32 tile goes straight from the
// accumulator registers into per-query candidate lists.
//
// Decomposition (one workgroup = 4 wavefronts = one CU, one wavefront per SIMD with the whole register file):
//   * a workgroup owns 128 queries (32 per wavefront) x one contiguous partition of the corpus;
//   * A (queries) is STATIONARY IN REGISTERS: lane (x = lane&31, h = lane>>5) keeps Q[x][8t + 4h + e] in
//     a[4t + e], t < NT = ceil(D/8): 4*NT VGPRs (192 for D = 384).  No LDS traffic for A at all;
//   * B (corpus) streams through LDS in tiles of 32 rows x D (49 KB at D = 384, double buffered) by LDS-DMA
//     (global_load_lds_dwordx4: no staging VGPRs, asynchronous): the DMA of tile t+1 is issued before tile t's MFMAs
//     and is complete at the single barrier per tile.  All 4 wavefronts consume the same tile against their own
//     queries; one ds_read_b128 feeds 4 MFMAs (a lane's 4 consecutive k's pair with a[4t..4t+3]); row pitch D+4
//     floats keeps the b128 reads bank-conflict free;
//   * epilogue per tile: D[i][j] for query i = (r&3)+8(r>>2)+4h and row j = lane&31 sits in acc[r]; distance ->
//     clamp -> compare with the query's current k-th best distance (32 floats per wavefront in LDS) -> only survivors
//     take the slow path into the query's sorted list in LDS (each list is owned by exactly one wavefront: no locks);
//     the epilogue of tile t-1 and the DMA issue of tile t+1 are interleaved INTO tile t's MFMA loop (one register /
//     one piece every few MFMAs) so that the matrix core never waits for them;
//   * one list per (query, partition) goes to HBM; vg_batch_merge_kernel rank-selects the final k per query.
//
// HBM traffic = corpus x (Q / 128) (each query group re-reads the corpus; the groups sharing a partition are placed
// on one XCD so the re-reads hit its L2); at ~100+ TF the kernel is MFMA-bound, not bandwidth-bound.
//
// Metrics: DOT, COSINE and L2 / squared L2 (row norms: vg_rownorm_kernel, cached per corpus; for L2 the GEMM is only
// the filter - survivors are re-evaluated with the reference formula).  L1 and the other element types are served
// by the single-query scan path (exact reference arithmetic) - see vg_api.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "vg_batch_common.h"

typedef float vgb_f32x16 __attribute__((ext_vector_type(16)));

#define VGB_THREADS 256
#define VGB_WAVES 4
#define VGB_QPW 32                      // queries per wavefront
#define VGB_QPB (VGB_WAVES * VGB_QPW)   // queries per workgroup
#define VGB_TILE 32                     // corpus rows per tile
#define VGB_MAX_K 32
#ifndef VGB_DUAL_ACC
#define VGB_DUAL_ACC 0                  // experiment: two independent accumulator chains per tile
#endif
#ifndef VGB_ABLATE
#define VGB_ABLATE 0                    // probe builds only: 1 = no threshold tests, 2 = no DMA after tile 0, 4 = no barrier
#endif

struct BatchArgs {
    const float *rows;        // N x stride_f floats
    const float *queries;     // nq_pad x stride_f floats (zero padded to a multiple of 128 queries)
    uint64_t *cand;           // [nq_pad][npart][64] keys
    long long n_rows;
    long long stride_f;       // floats between rows (multiple of 4)
    int nq_pad;
    int nq_real;              // queries [nq_real, nq_pad) are padding: they never accept a candidate
    int npart;
    int k;
    int mode;                 // VGB_DOT / VGB_COS / VGB_L2
    int root;                 // L2 mode: 1 = L2 (sqrt), 0 = squared L2
    int tiles_per_part;
    const float *xnorm;       // cosine, L2: ||row|| for every row (vg_rownorm_kernel, cached by the corpus)
    // this launch covers tiles [tile_begin, tile_end) in npart partitions and writes lists part_base .. part_base+npart-1
    // of the npart_total lists each query owns in `cand`
    long long tile_begin, tile_end;
    int part_base, npart_total;
    const uint64_t *init_keys; // optional [nq_pad][64]: a query's k-th key over rows scanned EARLIER bounds its answer
};
enum { VGB_DOT = 0, VGB_COS = 1, VGB_L2 = 2 };

typedef float vgb_f32x4 __attribute__((ext_vector_type(4)));
#ifndef VGB_BPIPE
#define VGB_BPIPE 4                      // B-operand ring depth, in k-steps of 4 MFMAs
#endif

// (free functions: clang rejects asm operands that name captured variables inside a generic lambda)
template <int OFF>
__device__ __forceinline__ void vgb_lds_read128(vgb_f32x4 &dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void vgb_wait_lds(vgb_f32x4 &v) {        // v is usable once at most N later LDS reads are in flight
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}


template <int NT, int MODE>
__global__ __launch_bounds__(VGB_THREADS, 1) void vg_batch_kernel(BatchArgs a) {
    constexpr bool COS = (MODE == VGB_COS), L2M = (MODE == VGB_L2);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int PITCH = NT * 8 + 4;                         // floats per LDS tile row (16-byte pad: conflict-free b128)
    constexpr int TILE_FLOATS = VGB_TILE * PITCH;
    constexpr int PIECES = (NT * 2 + 63) / 64;                // 1-KiB DMA pieces per row
    float *tile0 = reinterpret_cast<float *>(smem);
    float *tile1 = tile0 + TILE_FLOATS;
    float *thr_lds = tile1 + TILE_FLOATS;                     // [4][32] current k-th best distance per query
    float *qn_lds = thr_lds + VGB_WAVES * VGB_QPW;            // [4][32] ||q|| (cosine)
    uint64_t *lists = reinterpret_cast<uint64_t *>(qn_lds + VGB_WAVES * VGB_QPW);   // [4][32][k]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31, h = lane >> 5;
    const int k = a.k;

    // block -> (query group, partition); the groups that share a partition sit on one XCD (block b runs on XCD b % 8)
    const int G = a.nq_pad / VGB_QPB;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int g = idx % G;
    const int part = (idx / G) * 8 + xcd;
    if (part >= a.npart) return;
    const int q0 = g * VGB_QPB + wave * VGB_QPW;              // this wavefront's first query

    // ---- A operand: this lane's slice of query (q0 + x), kept in registers for the whole kernel
    float areg[NT * 4];
    float qq_part = 0.0f;
    {
        const float *qrow = a.queries + (long long)(q0 + x) * a.stride_f;
        vgb_static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int kk = 8 * t + 4 * h;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < a.stride_f) v = *reinterpret_cast<const float4 *>(qrow + kk);
            areg[4 * t + 0] = v.x; areg[4 * t + 1] = v.y; areg[4 * t + 2] = v.z; areg[4 * t + 3] = v.w;
            qq_part = fmaf(v.x, v.x, qq_part); qq_part = fmaf(v.y, v.y, qq_part);
            qq_part = fmaf(v.z, v.z, qq_part); qq_part = fmaf(v.w, v.w, qq_part);
        });
    }
    float *thr_w = thr_lds + wave * VGB_QPW;
    float *qn_w = qn_lds + wave * VGB_QPW;
    uint64_t *wave_lists = lists + (size_t)wave * VGB_QPW * k;
    {
        // ||q|| of query (q0+x): the two halves of the k range live in lanes x and x+32
        const float qq_x = qq_part + __shfl_xor(qq_part, 32);
        if (h == 0) { qn_w[x] = L2M ? qq_x : sqrtf(qq_x); thr_w[x] = INFINITY; }     // L2 keeps |q|^2, cosine |q|
        for (int s = lane; s < VGB_QPW * k; s += 64) wave_lists[s] = VG_EMPTY_KEY;
    }
    // zero both tile buffers once: the k-padding columns [stride_f, NT*8) are never touched by the DMA
    for (int s = tid; s < 2 * TILE_FLOATS; s += VGB_THREADS) tile0[s] = 0.0f;
    __syncthreads();

    // ---- tile streaming by LDS-DMA: wavefront w moves rows w, w+4, ... of the tile, one 1-KiB piece per instruction
    const int chunks_per_row = (int)(a.stride_f / 4);
    const long long tile_first = a.tile_begin + (long long)part * a.tiles_per_part;
    const long long tile_last = min(tile_first + a.tiles_per_part, a.tile_end);
    // One 1-KiB DMA piece: piece index pc in [0, 8*PIECES) = (row slot i, piece p) of this wavefront's 8 rows.
    // LDS-DMA from inline asm: hipcc's waitcnt pass does not see it, so it cannot put "s_waitcnt vmcnt(0)" in front
    // of every ds_read of the CURRENT tile while the NEXT tile is in flight (it did with the builtin: 16 exposed HBM
    // round trips per tile).  Completion is waited for explicitly before the tile barrier.
    // Addressing is scalar: SGPR pair = row base, one VGPR per piece column = lane byte offset (computed once);
    // M0 = wave-uniform LDS byte address of the piece; lane i lands at M0 + 16*i; inactive lanes write nothing.
    const uint32_t n_rows32 = (uint32_t)a.n_rows;
    const unsigned long long stride_b = (unsigned long long)a.stride_f * 4ull;
    uint32_t lane_off[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) lane_off[p] = (uint32_t)(p * 64 + lane) * 16u;
    uint64_t piece_mask[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) piece_mask[p] = __ballot(p * 64 + lane < chunks_per_row);
    const uint32_t lds_tile0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)tile0;
    auto dma_piece = [&](uint32_t tile32, int buf, int pc) {
        const int i = pc / PIECES, p = pc - i * PIECES;
        const int rr = wave + i * VGB_WAVES;
        uint32_t grow = tile32 * VGB_TILE + (uint32_t)rr;                      // rows past the end are masked later
        grow = grow < n_rows32 ? grow : n_rows32 - 1u;
        const uint8_t *sbase = reinterpret_cast<const uint8_t *>(a.rows) + (unsigned long long)grow * stride_b;
        const uint32_t lds_dst = lds_tile0 + (uint32_t)((buf * TILE_FLOATS + rr * PITCH + p * 256) * 4);
        // lanes past the end of the row are switched off by EXEC inside the asm (no branch: the k loop stays ONE
        // basic block, so the scheduler can keep the B-operand ds_reads several MFMAs ahead of their use)
        uint32_t keep;
        uint64_t keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_and_b64 exec, exec, %5\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(keep_exec) : "v"(lane_off[p]), "s"(sbase), "s"(lds_dst), "s"(piece_mask[p]) : "memory", "scc");
    };
    constexpr int NPIECE = (VGB_TILE / VGB_WAVES) * PIECES;

    // Per-register state, all in VGPRs (register r of lane (x, h) belongs to query qi(r, h) = (r&3) + 8*(r>>2) + 4*h):
    //   thr_reg[r]  current k-th best distance of that query (+Inf until its list is full)
    //   qn_reg[r]   ||q|| (cosine)
    //   gate[r]     in-loop gate on the RAW accumulator, a superset of "distance <= thr":
    //               dot:    d = -acc <= thr  <=>  acc >= -thr (minus a hair for the 8-eps clamp)
    //               cosine: 1 - acc/(|q||x|) <= thr  <=>  acc >= (1-thr)|q| * |x|; with G = (1-thr)|q| the test is
    //                       acc >= |x| * (G - 1e-5|G| - 1e-6) - 1e-30 (slack for the product / division roundings);
    //                       the |x|-independent factor is what gate[r] holds, one FMA per register remains in the loop.
    // The exact distance, clamp, row bound and key comparison happen in reg_insert (rare).  The full test with its LDS
    // read and clamp cost 30% of the kernel when it ran 16 times per tile.
    //               L2:     d2 = |q|^2 + |x|^2 - 2 acc <= thr2  <=>  acc >= (|q|^2 + |x|^2 - thr2) / 2; the gate keeps
    //                       the |x|-independent part, |q|^2 * 0.4999 - thr2 / 2, and the loop adds |x|^2 * 0.4999 (the
    //                       1e-4 relative slack covers the f32 error of acc).  The norm identity is ONLY this filter:
    //                       a survivor's distance is re-evaluated as sum (q-x)^2 from HBM in reg_insert, because the
    //                       identity cancels catastrophically for near-duplicates and the bar is 1e-5 relative.
    float gate[16], thr_reg[16], qn_reg[16];
    const bool l2_root = a.root != 0;
    // "accept everything" (list not full, zero-norm query) is a huge negative FINITE gate: -Inf would turn into NaN
    // against a zero-norm row (-Inf * 0) and a NaN margin is indistinguishable from "no" in the max reduction below
    auto make_gate = [&](float thr, float qn) -> float {
        float g;
        if (COS) {
            const float G = (1.0f - thr) * qn;
            g = G - 1e-5f * fabsf(G) - 1e-6f;
        } else if (L2M) {
            const float thr2 = l2_root ? thr * thr : thr;
            g = fmaf(qn, 0.4999f, -0.5f * thr2) - 1e-6f;
        } else {
            g = -thr - 1e-6f;
        }
        return fmaxf(g, -3.0e38f);                 // also maps a NaN gate to "accept"
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
        thr_reg[r] = a.init_keys ? vgb_kth_distance(a.init_keys[(long long)(q0 + qi) * 64 + (k - 1)]) : INFINITY;
        qn_reg[r] = qn_w[qi];
        gate[r] = make_gate(thr_reg[r], qn_reg[r]);
        if (q0 + qi >= a.nq_real) { thr_reg[r] = -INFINITY; gate[r] = 3.0e38f; }      // padding: an all-zero query would
    }                                                                                    // tie every row at cosine 1.0
    // the gate of register r for a tile whose rows have norm term `xterm` (cosine: |x|; L2: 0.4999 |x|^2)
    auto reg_gate = [&](auto rc, float xterm) -> float {
        constexpr int r = decltype(rc)::value;
        return COS ? fmaf(gate[r], xterm, -1e-30f) : (L2M ? gate[r] + xterm : gate[r]);
    };
    // distance of ONE accumulator register: acc_r = <query qi(r,h), row x>   (dot / cosine)
    auto reg_distance = [&](auto rc, float acc_r, float xnorm) -> float {
        constexpr int r = decltype(rc)::value;
        float d;
        if (COS) d = vg_cosine_from_norms(acc_r, qn_reg[r], xnorm);
        else d = -acc_r;
        return vg_clamp(d);
    };
    // L2: the reference's own formula for ONE (query, row) pair, computed by the whole wavefront from HBM / L2
    // (wave-uniform arguments): lane-strided float4 loads, 4 f32 FMA partials, butterfly - the single-query kernel's
    // arithmetic.  ~1 us of latency per survivor; survivors are k*ln(n/k) per list.
    auto exact_l2 = [&](int q_uniform, long long row_uniform) -> float {
        const float4 *qp = reinterpret_cast<const float4 *>(a.queries + (long long)(q0 + q_uniform) * a.stride_f);
        const float4 *xp = reinterpret_cast<const float4 *>(a.rows + row_uniform * a.stride_f);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int c = lane; c < chunks_per_row; c += 64) {
            const float4 qv = qp[c], xv = xp[c];
            const float d0 = qv.x - xv.x, d1 = qv.y - xv.y, d2 = qv.z - xv.z, d3 = qv.w - xv.w;
            s0 = fmaf(d0, d0, s0); s1 = fmaf(d1, d1, s1); s2 = fmaf(d2, d2, s2); s3 = fmaf(d3, d3, s3);
        }
        const float s = vg_group_sum((s0 + s1) + (s2 + s3), 6);
        return vg_clamp(l2_root ? sqrtf(s) : s);
    };
    // slow path (outside the MFMA loop, rare once the lists have warmed up): insert this register's survivors
    auto reg_insert = [&](auto rc, float acc_r, long long row, float xnorm) {
        constexpr int r = decltype(rc)::value;
        const int q_lo = (r & 3) + 8 * (r >> 2);
        float d = 0.0f;
        bool pass;
        if (L2M) {
            // the gate again (per lane this time); the distance itself comes from exact_l2 below
            pass = (row < a.n_rows) && (acc_r >= gate[r] + xnorm * xnorm * 0.4999f);
        } else {
            d = reg_distance(rc, acc_r, xnorm);
            // strict: rows arrive in scan order, so a row that only TIES the k-th best (here or in the pre-pass,
            // whose rows all come earlier) has the larger position and loses; +Inf thresholds accept every finite d
            pass = (row < a.n_rows) && (d < thr_reg[r]);
        }
        unsigned long long m = __ballot(pass);
        uint64_t key = vg_make_key(d, (uint32_t)row);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int hh = src >> 5;
            uint64_t *list = wave_lists + (q_lo + 4 * hh) * k;
            uint64_t c;
            if (L2M) {
                const long long row_u = __builtin_amdgcn_readlane((int)row, src);          // rows < 2^32 per shard, and
                const float thr_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thr_reg[r]), src));
                const float de = exact_l2(q_lo + 4 * hh, (long long)(uint32_t)row_u);       // positive as u32
                if (!(de < thr_u)) continue;
                c = vg_make_key(de, (uint32_t)row_u);
            } else {
                c = vg_readlane64(key, src);
            }
            // one LDS round trip per candidate: a key that no longer beats the tail simply changes nothing below
            const float nt = vgb_kth_distance(vgb_list_insert(list, k, lane, c));
            if (h == hh) {
                thr_reg[r] = fminf(nt, thr_reg[r]);          // never loosens (a pre-pass bound outlives a not-yet-full list)
                gate[r] = make_gate(thr_reg[r], qn_reg[r]);
            }
        }
    };

    if (tile_first < tile_last) {
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) dma_piece((uint32_t)tile_first, 0, pc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // land this wavefront's DMA pieces ...
    __syncthreads();                                          // ... and everybody else's

    // Per tile: the k loop holds NOTHING but MFMAs, the B-operand LDS reads and the DMA issue of the next tile.  On
    // this machine ONE extra VALU issue between two MFMAs of the same accumulator chain costs ~43 cycles (the
    // back-to-back accumulator forwarding is lost; MI355X_MICROARCH.md, per-instruction constants) - gating the
    // previous tile's 16 score registers "in the shadow" of the loop, one register every few MFMAs, opened ~20 such
    // gaps per tile and cost 7% of the matrix pipe.  The scores are gated at the tile boundary instead: one block of
    // 16 v_sub + 8 v_max3 behind the chain's drain, one ballot, and (rarely) the inserts - then the barrier.
    unsigned dbg_tiles = 0, dbg_pend = 0, dbg_regs = 0;        // probe builds (VGB_ABLATE & 32) only
    // B operand pipeline: ds_read_b128 issued from inline asm BP steps (= 4*BP MFMAs) ahead of its use into a ring
    // of BP register quads, with an exact "s_waitcnt lgkmcnt(n)" in front of the consumer.  Left to the compiler
    // the dot variant re-used ONE register quad: read, lgkmcnt(0), 4 MFMAs, read ... - an exposed LDS round trip
    // every 4-8 MFMAs with a single wavefront per SIMD and nothing else to issue (22% of the matrix pipe idle).
    constexpr int BP = VGB_BPIPE < NT ? VGB_BPIPE : NT;
    vgb_f32x4 bq[BP];
    for (long long tile = tile_first; tile < tile_last; ++tile) {
        const int cur_buf = (int)((tile - tile_first) & 1);
        const float *cur = cur_buf ? tile1 : tile0;
        const uint32_t tile_next = (uint32_t)min(tile + 1, tile_last - 1);   // the last iteration re-fetches its own tile: harmless
        const long long row_cur = tile * VGB_TILE + x;
        // cosine / L2: ||x|| of this lane's row comes from the corpus' cached norm vector (one global load per tile,
        // consumed after the k loop); accumulating it from the B reads cost 4 VALU FMAs per k-step and ~20 TFLOP/s
        float xnorm_cur = 0.0f;
        if (COS || L2M) xnorm_cur = a.xnorm[row_cur < a.n_rows ? row_cur : a.n_rows - 1];

        vgb_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float *brow = cur + x * PITCH + 4 * h;
        const uint32_t baddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)brow;
        vgb_static_for<0, BP>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            vgb_lds_read128<32 * t>(bq[t], baddr);
        });
        // compile-time unrolled k loop (a template recursion: the plain "#pragma unroll" gave up on a body this large
        // and put areg[] in scratch memory)
        vgb_static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int in_flight_after = (NT - 1 - t) < (BP - 1) ? (NT - 1 - t) : (BP - 1);
            vgb_wait_lds<in_flight_after>(bq[t % BP]);
            const vgb_f32x4 b = bq[t % BP];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * t + 0], b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * t + 1], b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * t + 2], b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * t + 3], b.w, acc, 0, 0, 0);
            if constexpr (t + BP < NT) vgb_lds_read128<32 * (t + BP)>(bq[t % BP], baddr);
            // DMA pieces of the next tile go out during the FIRST THIRD of the k loop (the rest of the MFMAs cover
            // their HBM latency)
            constexpr int NTD = (NT + 2) / 3;
            constexpr int pc_lo = (t >= NTD) ? NPIECE : (t * NPIECE + NTD - 1) / NTD;
            constexpr int pc_hi = (t >= NTD) ? NPIECE : (t + 1 == NTD ? NPIECE : ((t + 1) * NPIECE + NTD - 1) / NTD);
            if (!(VGB_ABLATE & 2)) vgb_static_for<pc_lo, pc_hi>([&](auto pcc) { dma_piece(tile_next, cur_buf ^ 1, decltype(pcc)::value); });
        });

        // ---- tile boundary: gate the 16 score registers of THIS tile.  VALU only: margin = score - gate, max-reduced;
        // ONE ballot decides whether anything needs the slow path.  A NaN margin (NaN score = NaN distance, never a
        // result) is ignored by the max.
        const float xterm = COS ? xnorm_cur : (L2M ? xnorm_cur * xnorm_cur * 0.4999f : 0.0f);
        float margin = -INFINITY;
        if (!(VGB_ABLATE & 1)) {
            vgb_static_for<0, 16>([&](auto rc) { margin = fmaxf(margin, acc[decltype(rc)::value] - reg_gate(rc, xterm)); });
        }
        unsigned pend = 0;
        if (__ballot(margin >= 0.0f)) {
            // which registers: the same test per register (rare path)
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                pend |= __ballot(acc[r] >= reg_gate(rc, xterm)) ? (1u << r) : 0u;
            });
        }
        if (VGB_ABLATE & 8) { asm volatile("" ::"s"(pend)); pend = 0; }          // probe: gates computed, never acted on
        if (VGB_ABLATE & 64) pend &= (unsigned)(k >> 10);                          // probe: slow path compiled in, never run
        if (VGB_ABLATE & 32) { dbg_tiles++; dbg_pend += pend ? 1 : 0; dbg_regs += __builtin_popcount(pend); }
        if (pend) {
            vgb_static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if (pend & (1u << r)) reg_insert(rc, acc[r], row_cur, xnorm_cur);
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of tile t+1 have landed
        if (!(VGB_ABLATE & 4)) __syncthreads();               // tile t consumed by all, tile t+1 landed for all
    }
    if ((VGB_ABLATE & 32) && lane == 0 && blockIdx.x < 2)
        printf("block %d wave %d: tiles %u with-pend %u flagged-regs %u\n", blockIdx.x, wave, dbg_tiles, dbg_pend, dbg_regs);

    // ---- publish: [query][part][64] (a query's npart lists are contiguous for the merge)
    for (int s = lane; s < VGB_QPW * 64; s += 64) {
        const int qi = s >> 6, slot = s & 63;
        a.cand[((long long)(q0 + qi) * a.npart_total + a.part_base + part) * 64 + slot] = (slot < k) ? wave_lists[qi * k + slot] : VG_EMPTY_KEY;
    }
}

// ||row|| for rows [row0, row0 + n): 16 lanes per row, float4 loads, f32 FMA partials + butterfly (the batched cosine
// path is a <= 1e-5 path; the bit-exact reference order lives in the single-query kernels).  HBM-bound, run once per
// appended row: the corpus caches the vector.
__global__ __launch_bounds__(256) void vg_rownorm_kernel(const float *rows, long long row0, long long n, long long stride_f,
                                                         float *out) {
    const int sub = threadIdx.x & 15;
    const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ngroups = ((long long)gridDim.x * blockDim.x) >> 4;
    const int nch = (int)(stride_f / 4);
    for (long long r = group; r < n; r += ngroups) {
        const float4 *p = reinterpret_cast<const float4 *>(rows + (row0 + r) * stride_f);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int c = sub; c < nch; c += 16) {
            const float4 v = p[c];
            s0 = fmaf(v.x, v.x, s0); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3);
        }
        float s = (s0 + s1) + (s2 + s3);
        s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
        if (sub == 0) out[row0 + r] = sqrtf(s);
    }
}

extern "C" int vg_rownorm_launch(const float *dev_rows, long long row0, long long n, long long stride_bytes, float *dev_out,
                                 hipStream_t stream) {
    if (n <= 0) return 0;
    long long blocks = (n * 16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(vg_rownorm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dev_rows, row0, n, stride_bytes / 4, dev_out);
    return (int)hipGetLastError();
}

// per query: its npart sorted lists (contiguous in `cand`) -> final k (ascending, EMPTY padded to 64).
// One workgroup per query, same parallel rank-select as the single-query path.
__global__ __launch_bounds__(256) void vg_batch_merge_kernel(const uint64_t *cand, int nq_pad, int lists_per_query,
                                                             int npart, int k, uint64_t *out_keys) {
    __shared__ __attribute__((aligned(16))) uint8_t scratch[VG_SEL_SCRATCH_BYTES];
    const int q = blockIdx.x;
    vg_select_lists(cand + (long long)q * lists_per_query * 64, npart, k, out_keys + (long long)q * 64, scratch);
}

extern "C" int vg_batch_merge_launch(const uint64_t *dev_cand, int nq_pad, int lists_per_query, int npart, int k,
                                     uint64_t *dev_out_keys, hipStream_t stream) {
    hipLaunchKernelGGL(vg_batch_merge_kernel, dim3((unsigned)nq_pad), dim3(256), 0, stream, dev_cand, nq_pad, lists_per_query,
                       npart, k, dev_out_keys);
    return (int)hipGetLastError();
}

template <int NT, int MODE>
static int launch_nt_mode(const BatchArgs &a, int blocks, size_t smem, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vg_batch_kernel<NT, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((vg_batch_kernel<NT, MODE>), dim3((unsigned)blocks), dim3(VGB_THREADS), smem, stream, a);
    return (int)hipGetLastError();
}
template <int NT>
static int launch_nt(const BatchArgs &a, int blocks, size_t smem, hipStream_t stream) {
    if (a.mode == VGB_COS) return launch_nt_mode<NT, VGB_COS>(a, blocks, smem, stream);
    if (a.mode == VGB_L2) return launch_nt_mode<NT, VGB_L2>(a, blocks, smem, stream);
    return launch_nt_mode<NT, VGB_DOT>(a, blocks, smem, stream);
}

// LDS bytes of the batch kernel for a row of `stride_bytes` and k; 0 if the shape is not served
extern "C" size_t vg_batch_lds_bytes(long long stride_bytes, int k) {
    const int nt = (int)((stride_bytes / 4 + 7) / 8);
    int NT;
    if (nt <= 16) NT = 16; else if (nt <= 32) NT = 32; else if (nt <= 48) NT = 48; else if (nt <= 64) NT = 64;
    else return 0;
    if (k < 1 || k > VGB_MAX_K) return 0;
    const size_t b = (size_t)2 * VGB_TILE * (NT * 8 + 4) * 4 + (size_t)2 * VGB_WAVES * VGB_QPW * 4 +
                     (size_t)VGB_WAVES * VGB_QPW * k * 8;
    return b <= 160 * 1024 ? b : 0;
}

// How many candidate lists a query owns in `dev_cand` for this shape (the caller sizes the buffer with it): npart for
// a single pass, 2 * npart with the pre-pass (see vg_batch_launch).
extern "C" int vg_batch_prepass_tiles(long long n_rows, int npart) {
    const int denom = vg_sw(SW_VG_BATCH_PREPASS, 64);                  // pre-pass over 1/denom of the corpus; 0 = off
    const long long ntiles = (n_rows + VGB_TILE - 1) / VGB_TILE;
    if (denom <= 0 || ntiles < 65536 || 2 * npart > VG_SEL_MAX_HEADS) return 0;   // < 2M rows: a single pass
    long long t = ntiles / denom;
    t = ((t + npart - 1) / npart) * npart;                        // whole partitions
    return (int)t;
}
extern "C" int vg_batch_lists_per_query(long long n_rows, int npart) {
    return vg_batch_prepass_tiles(n_rows, npart) > 0 ? 2 * npart : npart;
}

// Host launcher.  Returns 0 on success, -1 if the shape is not served by this kernel (caller falls back to the
// single-query path), a hipError_t otherwise.  dev_cand: nq_pad x vg_batch_lists_per_query() x 64 keys;
// dev_out_keys: nq_pad x 64 keys.  A lives in 4*NT VGPRs per lane, so rows up to 512 floats are served.
//
// Large corpora run in TWO passes.  Every (query, partition) list costs k*ln(n/k) inserts (193 at C5), and an insert
// stalls its wavefront - and through the per-tile barrier its workgroup - with the matrix pipe idle.  Pass 1 scans
// the first 1/64 of the corpus and merges; each query's k-th best there is an upper bound on its final k-th best, so
// pass 2 starts every list at that threshold instead of +Inf: ~k*(1 + ln(rows_per_partition / rows_in_pass_1)) inserts
// per list.  The final merge takes the lists of both passes.
extern "C" int vg_batch_launch(const float *dev_rows, long long n_rows, long long stride_bytes,
                               const float *dev_queries, int nq_pad, int nq_real, int k, int mode, int root,
                               const float *dev_xnorm, uint64_t *dev_cand, int npart, int tiles_per_part,
                               uint64_t *dev_out_keys, hipStream_t stream) {
    const size_t smem = vg_batch_lds_bytes(stride_bytes, k);
    if (!smem || nq_pad % VGB_QPB != 0 || npart < 1 || npart > VG_SEL_MAX_HEADS || n_rows < 1) return -1;
    if (mode < VGB_DOT || mode > VGB_L2 || (mode != VGB_DOT && !dev_xnorm)) return -1;
    BatchArgs a;
    a.rows = dev_rows; a.queries = dev_queries; a.cand = dev_cand; a.n_rows = n_rows;
    a.stride_f = stride_bytes / 4; a.nq_pad = nq_pad; a.nq_real = nq_real; a.npart = npart; a.k = k; a.mode = mode; a.root = root;
    a.xnorm = dev_xnorm;
    const int nt = (int)((a.stride_f + 7) / 8);
    const int G = nq_pad / VGB_QPB;
    const int blocks = G * ((npart + 7) / 8) * 8;
    const long long ntiles = (n_rows + VGB_TILE - 1) / VGB_TILE;
    auto launch = [&](const BatchArgs &b) -> int {
        if (nt <= 16) return launch_nt<16>(b, blocks, smem, stream);
        if (nt <= 32) return launch_nt<32>(b, blocks, smem, stream);
        if (nt <= 48) return launch_nt<48>(b, blocks, smem, stream);
        return launch_nt<64>(b, blocks, smem, stream);
    };
    const long long pre = vg_batch_prepass_tiles(n_rows, npart);
    int rc;
    if (pre > 0) {
        a.npart_total = 2 * npart;
        a.tile_begin = 0; a.tile_end = pre; a.tiles_per_part = (int)(pre / npart); a.part_base = 0; a.init_keys = nullptr;
        if ((rc = launch(a)) != 0) return rc;
        hipLaunchKernelGGL(vg_batch_merge_kernel, dim3((unsigned)nq_pad), dim3(256), 0, stream, (const uint64_t *)dev_cand,
                           nq_pad, a.npart_total, npart, k, dev_out_keys);
        a.tile_begin = pre; a.tile_end = ntiles; a.tiles_per_part = (int)((ntiles - pre + npart - 1) / npart);
        a.part_base = npart; a.init_keys = dev_out_keys;
        if ((rc = launch(a)) != 0) return rc;
    } else {
        a.npart_total = npart;
        a.tile_begin = 0; a.tile_end = ntiles; a.tiles_per_part = tiles_per_part; a.part_base = 0; a.init_keys = nullptr;
        if ((rc = launch(a)) != 0) return rc;
    }
    hipLaunchKernelGGL(vg_batch_merge_kernel, dim3((unsigned)nq_pad), dim3(256), 0, stream, (const uint64_t *)dev_cand,
                       nq_pad, a.npart_total, a.npart_total, k, dev_out_keys);
    return (int)hipGetLastError();
}
