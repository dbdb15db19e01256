// vg_batch_common.h - pieces shared by the batched kernels (vg_batch.hip: f32, vg_batch_i8.hip: uint8 / int8).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vg_lists.h"

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
