// vg_device.h - device-side building blocks of the gfx950 scan: packed (distance, position) keys,
// wavefront-wide sorted candidate list, per-(type, metric) accumulators and epilogues.
//
// CDNA4 only: 64-lane wavefronts are assumed everywhere (ballots are 64-bit, lists are one slot per lane).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VG_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define VG_WAVE 64
#define VG_BLOCK 1024                // 16 wavefronts per workgroup = one workgroup per CU at ~96 VGPRs; the 16 wave
                                     // lists merge in LDS, so only ONE candidate list per CU reaches HBM
#define VG_WAVES_PER_BLOCK (VG_BLOCK / VG_WAVE)
#define VG_MAX_FUSED_K 64            // one list slot per lane

// element types / metrics: numbering of the reference (distance-cpu.h:36-58)
enum { T_F32 = 1, T_F16 = 2, T_BF16 = 3, T_U8 = 4, T_I8 = 5 };
enum { M_L2 = 1, M_SQL2 = 2, M_COS = 3, M_DOT = 4, M_L1 = 5 };
// accumulation kinds (L2 and squared-L2 share one; the root is an epilogue flag)
enum { A_L2 = 0, A_COS = 1, A_DOT = 2, A_L1 = 3, A_COSN = 4 /* f16/bf16 cosine with the row norms read from a cached vector */ };

struct ScanArgs {
    const uint8_t *rows;       // N x stride bytes, 16-byte-multiple stride, zero padded
    const uint8_t *query;      // nch * 16 bytes, zero padded (device)
    uint64_t *cand;            // [gridDim.x][64] block candidate lists (top-k mode)
    float *out_dist;           // [n_rows] all distances (store mode) or nullptr
    long long n_rows;
    long long stride;          // bytes between rows
    int nch;                   // 16-byte chunks per row
    int lpr_log2;              // log2(lanes cooperating on one row)
    int k;                     // <= 64 in top-k mode
    int root;                  // 1: L2 (sqrt), 0: squared L2
    int dim;                   // elements per row (for the special-value slow paths)
    const float *row_nn;       // A_COSN: (float) sum x^2 per row (vg_half_rownorm_kernel); nullptr otherwise
    int store_lds_off;         // store mode: byte offset in dynamic LDS of the per-wavefront staging areas
    // ---- tie_order = reference support (vg_reforder.hip), all optional
                               //   out_dist != nullptr with k > 0: top-k mode that ALSO writes every row's distance (the replay's prefix pass)
    const uint64_t *init_keys; // the k best of a scan over the rows in front (64 keys) or nullptr: key k-1 is every list's start threshold
    unsigned long long *emit;  // [count | emit_cap pairs]: every row a list accepts whose distance is strictly below init_keys' k-th
                               //   distance is appended as (position << 32 | float bits) - a superset of the rows that can enter the
                               //   reference's k slots (it is below the k-th best of SOME earlier rows, hence possibly of all of them)
    unsigned emit_cap;
    unsigned long long *emit_reset;   // zeroed by this launch's first thread (the prefix pass resets the counter of the pass behind it)
    int order;                 // top-k mode: 0 = grid-stride batches (the whole chip sweeps one contiguous window), 1 = every workgroup owns a
                               //   contiguous run of batches, its 16 wavefronts striding inside it (one CU stays on one page for many iterations)
};

// ------------------------------------------------------------------------------------------ keys

// order-preserving map float -> uint32 (ascending floats -> ascending unsigned)
__host__ __device__ inline uint32_t vg_f32_sortable(float d) {
    uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = __float_as_uint(d);
#else
    __builtin_memcpy(&b, &d, 4);
#endif
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float vg_sortable_f32(uint32_t s) {
    uint32_t b = s ^ ((s >> 31) ? 0x80000000u : 0xFFFFFFFFu);
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f; __builtin_memcpy(&f, &b, 4); return f;
#endif
}
__host__ __device__ inline uint64_t vg_make_key(float d, uint32_t pos) {
    return ((uint64_t)vg_f32_sortable(d) << 32) | (uint64_t)pos;
}

// ------------------------------------------------------------------------------------------ wave helpers

__device__ inline uint64_t vg_readlane64(uint64_t v, int lane_uniform) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane_uniform);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane_uniform);
    return ((uint64_t)hi << 32) | lo;
}

// Sorted candidate list spread over the wavefront: lane i holds the i-th smallest key, `thr` (wave-uniform)
// is the key in slot k-1, i.e. the current k-th best.  Insert = one shift-up + selects (no loop, no LDS).
// whole-wavefront shift by one lane (lane i <- lane i-1, lane 0 <- 0) as two DPP moves (v_mov_b32_dpp wave_shr:1):
// no LDS crossbar round trip, unlike __shfl_up (ds_bpermute_b32)
__device__ inline uint64_t vg_wave_shr1(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}

// thr_cap: a start threshold from elsewhere (the k-th best of rows scanned by an earlier pass) keeps bounding thr while the list
// itself holds fewer than k keys
__device__ inline void vg_list_insert(uint64_t &mine, uint64_t &thr, uint64_t c, int lane, int k, uint64_t thr_cap = VG_EMPTY_KEY) {
    const uint64_t prev = vg_wave_shr1(mine);        // lane 0 sees 0, which is never > c
    const bool gt = mine > c;
    const bool pgt = prev > c;
    mine = gt ? (pgt ? prev : c) : mine;
    const uint64_t kth = vg_readlane64(mine, k - 1);
    thr = kth < thr_cap ? kth : thr_cap;
}

// tie_order = reference: one (position, distance) pair appended to the candidate stream (wave-uniform arguments, one lane writes)
__device__ inline void vg_emit_pair(unsigned long long *emit, unsigned cap, uint64_t key, int lane) {
    if (lane == 0) {
        const unsigned long long slot = atomicAdd(emit, 1ull);
        if (slot < cap) emit[1 + slot] = ((key & 0xFFFFFFFFull) << 32) | (unsigned long long)__float_as_uint(vg_sortable_f32((uint32_t)(key >> 32)));
    }
}

// Offer one candidate per lane (valid lanes only).  Expected cost ~0 once thr has tightened.
__device__ inline void vg_list_offer(uint64_t key, bool valid, uint64_t &mine, uint64_t &thr, int lane, int k) {
    unsigned long long m = __ballot(valid && key < thr);
    while (m) {
        int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        uint64_t c = vg_readlane64(key, src);
        if (c < thr) vg_list_insert(mine, thr, c, lane, k);
    }
}
// The same with a start threshold and the reference-order candidate stream (vg_reforder.hip).  The three values involved are
// needed only when a list accepts a key - about once per wavefront and scan - so they live in LDS, not in the scalar registers the
// streaming loop is short of (every scan kernel sits at the 106-SGPR limit; holding them there spilled SGPRs into VGPR lanes).
struct VgListExtras {
    uint64_t thr_cap;            // key: the k-th best of the rows an earlier pass scanned (EMPTY: none) - bounds thr while a list is short
    uint64_t emit_below;         // accepted keys below this are emitted (0: nothing is)
    unsigned long long *emit;    // [count | emit_cap pairs] or nullptr
    unsigned long long emit_cap;
};
__device__ inline uint64_t vg_uniform64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// ex points into LDS (filled before the workgroup's first barrier)
__device__ inline void vg_list_offer_ex(uint64_t key, bool valid, uint64_t &mine, uint64_t &thr, int lane, int k, const VgListExtras *ex) {
    unsigned long long m = __ballot(valid && key < thr);
    while (m) {
        int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        uint64_t c = vg_readlane64(key, src);
        if (c < thr) {
            vg_list_insert(mine, thr, c, lane, k, vg_uniform64(ex->thr_cap));
            if (c < vg_uniform64(ex->emit_below))
                vg_emit_pair(reinterpret_cast<unsigned long long *>(vg_uniform64((uint64_t)ex->emit)), (unsigned)ex->emit_cap, c, lane);
        }
    }
}
// fills the LDS block from a launch's arguments (one thread; the caller's next barrier publishes it) and returns the start threshold
__device__ inline uint64_t vg_list_extras_init(VgListExtras *ex, const uint64_t *init_keys, int k, unsigned long long *emit, unsigned emit_cap) {
    uint64_t thr_cap = VG_EMPTY_KEY, emit_below = emit ? VG_EMPTY_KEY : 0ull;    // no pass in front: everything a list accepts can enter the slots
    if (init_keys) {
        const uint64_t kk = init_keys[k - 1];
        if (kk != VG_EMPTY_KEY) {                            // <= its distance enters a list, < is emitted
            thr_cap = kk | 0xFFFFFFFFull;
            if (emit) emit_below = kk & 0xFFFFFFFF00000000ull;
        }
    }
    if (threadIdx.x == 0) { ex->thr_cap = thr_cap; ex->emit_below = emit_below; ex->emit = emit; ex->emit_cap = emit_cap; }
    return thr_cap;
}

// Butterfly sum over the 2^lpr_log2 lanes that share a row; every lane of the group ends with the (bitwise
// identical) total.  Steps 1/2/4/8 stay inside a 16-lane DPP row and are single DPP moves feeding the add
// (quad_perm, quad_perm, row_half_mirror, row_mirror) - no ds_bpermute round trip through the LDS crossbar, whose
// ~100-cycle dependent latency per step made the 2-byte types latency-bound; only the 32- and 64-lane steps
// (rows of >= 512 bytes per load) use the crossbar.
#define VG_DPP_QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define VG_DPP_ROW_MIRROR 0x140
#define VG_DPP_ROW_HALF_MIRROR 0x141

template <int CTRL> __device__ inline uint32_t vg_dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ inline float vg_dpp(float v) { return __uint_as_float(vg_dpp_u32<CTRL>(__float_as_uint(v))); }
template <int CTRL> __device__ inline uint32_t vg_dpp(uint32_t v) { return vg_dpp_u32<CTRL>(v); }
template <int CTRL> __device__ inline double vg_dpp(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint64_t r = ((uint64_t)vg_dpp_u32<CTRL>((uint32_t)(b >> 32)) << 32) | vg_dpp_u32<CTRL>((uint32_t)b);
    return __longlong_as_double((long long)r);
}

template <typename T>
__device__ inline T vg_group_sum(T v, int lpr_log2) {
    if (lpr_log2 > 0) v += vg_dpp<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(v);
    if (lpr_log2 > 1) v += vg_dpp<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(v);
    if (lpr_log2 > 2) v += vg_dpp<VG_DPP_ROW_HALF_MIRROR>(v);
    if (lpr_log2 > 3) v += vg_dpp<VG_DPP_ROW_MIRROR>(v);
    if (lpr_log2 > 4) v += __shfl_xor(v, 16);
    if (lpr_log2 > 5) v += __shfl_xor(v, 32);
    return v;
}

// "does any lane of my group have flag != 0": one ballot + a per-lane shift/mask instead of a shuffle butterfly
__device__ inline uint32_t vg_group_or(uint32_t flag, int lpr_log2) {
    const unsigned long long m = __ballot(flag != 0);
    const int lane = (int)(threadIdx.x & (VG_WAVE - 1));
    const int base = lane & ~((1 << lpr_log2) - 1);
    const unsigned long long gm = (lpr_log2 >= 6) ? ~0ull : ((1ull << (1 << lpr_log2)) - 1ull);
    return ((m >> base) & gm) != 0ull ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------ epilogue math
// Each operation below must round exactly once, like the reference's scalar epilogues: correctly rounded
// sqrt / divide, no contraction of a*b+c.

#pragma clang fp contract(off)

// nearly_zero_float32 clamp, sqlite-vector.c:994-996 applied at :2099 / :2141
__device__ inline float vg_clamp(float d) { return (fabsf(d) <= 8.0f * 1.1920928955078125e-7f) ? 0.0f : d; }

// 1 - dot / (na * nb) with the zero-norm rule (distance-avx2.c:153-162, :744-753, :941-950; distance-cpu.c:103-109)
__device__ inline float vg_cosine_from_norms(float dot, float na, float nb) {
    if (na == 0.0f || nb == 0.0f) return 1.0f;
    float den = na * nb;
    float cs = __fdiv_rn(dot, den);
    return 1.0f - cs;
}
