// vg_batch_common.h - pieces shared by the batched kernels (vg_batch.hip: f32, vg_batch_i8.hip: uint8 / int8).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vg_lists.h"
#include "vg_switches.h"

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)  (a plain "#pragma unroll" gives up on the
// kernels' k loops and demotes the register-resident operands to scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void vgb_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        vgb_static_for<I + 1, N>(f);
    }
}

// One candidate key into a sorted per-query list of k keys in LDS (lane i of the wavefront owns slot i, k <= 64), one
// LDS round trip: a key that does not beat the tail changes nothing.  `c` is wave-uniform.  Returns the new k-th key.
__device__ __forceinline__ uint64_t vgb_list_insert(uint64_t *list, int k, int lane, uint64_t c) {
    uint64_t mine = (lane < k) ? list[lane] : 0ull;
    const uint64_t prev = vg_wave_shr1(mine);
    mine = (mine > c) ? ((prev > c) ? prev : c) : mine;
    if (lane < k) list[lane] = mine;
    return vg_readlane64(mine, k - 1);
}

// the distance a list's k-th key stands for; a list that is not full yet accepts everything
__device__ __forceinline__ float vgb_kth_distance(uint64_t kth) {
    return (kth == VG_EMPTY_KEY) ? INFINITY : vg_sortable_f32((uint32_t)(kth >> 32));
}

// ---- STAGED REAL PASSES (host side).  A two-pass launch starts every list of the real pass at the query's k-th best of a
// pre-pass over 1/32 of the rows: ~k * 32 rows per query still beat that threshold in the real pass (655k slow-path entries
// for 1024 queries x 10M rows), and each of them holds its workgroup at a tile barrier.  The real pass therefore runs in
// STAGES over growing row ranges [0, 2P), [2P, 4P), [4P, 8P) ... (P = the pre-pass rows): the lists are merged after every
// stage and the next one starts from the query's k-th best over ALL rows scanned so far - a stage that doubles the scanned
// range lets ~k rows per query through, ~6k per query in all instead of 32k.  From the second stage on, partition 0 of a
// query group seeds its lists with the merged keys (they stand for rows of EARLIER stages, which no later stage meets again),
// so every stage writes - and every merge reads - the same npart lists per query.
// bounds[0 .. n]: stage i scans tiles [bounds[i], bounds[i+1]); returns n.  growth_pct: VG_BATCH_STAGES (200 = doubling,
// 0 = one real pass over everything), else the caller's default.
// Round 3 re-measured both knobs on the kernels as they are now (tile-major copies, tile-minimum pre-pass, staged passes;
// tools/batch_knob_sweep.py, profiles/r4h_batch_knob_sweeps.jsonl): the pre-pass wants to be SMALL - 1/512 of the rows instead of
// 1/32 - because the staged passes do the warming-up more cheaply than a longer pre-pass does: 1024 x 10M, f32 through the bf16
// filter 10.54 -> 8.95 ms, f16 dot 9.92 -> 8.68, bf16 cosine 10.03 -> 8.43, uint8 cosine 9.01 -> 7.61, uint8 dot 7.61 -> 6.57,
// int8 L2 7.70 -> 6.61; 256 queries 3.12 -> 2.72 (f32) / 2.70 -> 2.24 (uint8).  Growth: doubling for the half-precision kernel
// with several query groups (its exact evaluations are the expensive part), x4 otherwise (fewer launches).
#include <cstdlib>
#define VGB_PREPASS_DENOM_DEFAULT 512
static inline int vgb_stage_bounds(long long ntiles, long long pre_tiles, long long *bounds, int max_stages, int default_growth_pct = 400) {
    const int growth_pct = vg_sw(SW_VG_BATCH_STAGES, default_growth_pct);
    int n = 0;
    bounds[0] = 0;
    if (pre_tiles > 0 && growth_pct > 100) {
        long long b = 2 * pre_tiles;
        while (b < ntiles && n + 2 < max_stages && ntiles - b > b / 4) {      // (no sliver at the end)
            bounds[++n] = b;
            b = b * growth_pct / 100;
        }
    }
    bounds[++n] = ntiles;
    return n;
}
