// vg_lists.h - workgroup-parallel selection of the k smallest keys out of L sorted candidate lists.
//
// Used twice per query: inside the scan kernel (the 16 wave lists of a workgroup -> one list per CU) and by
// vg_merge_kernel (one list per CU -> the final k).  Serial sorted-list inserts cost ~0.1-0.25 us each on one
// wavefront (probe: tools/merge_probe.hip), which made these merges 10-25 us; this version is a few barriers:
//
//   1. heads:     take the first m = ceil(k / L) keys of every list (L*m >= k of them).  tau = the k-th smallest head
//                 is an upper bound on the k-th smallest key overall (those k heads are all <= tau), and a tight one;
//                 every head's rank is computed by its own thread (H^2 / threads comparisons, H <= 256).
//   2. survivors: keys <= tau are the only possible answers.  Lists are sorted and keys unique, so at most
//                 k*k/m + k (<= VG_SURV_CAP) keys survive - typically ~k.  They are compacted into LDS with one
//                 LDS atomic each.
//   3. rank:      every survivor's rank among the survivors is its position in the output; ranks < k are written.
//
// Keys are unique (low 32 bits = scan position), so ranks are a permutation and the output is strictly ascending.
#pragma once

#include "vg_device.h"

#define VG_SEL_MAX_HEADS 256
#define VG_SURV_CAP (64 * 64 + 64)
// scratch bytes needed in LDS by vg_select_lists (heads + survivors + 2 words)
#define VG_SEL_SCRATCH_BYTES ((VG_SEL_MAX_HEADS + VG_SURV_CAP) * 8 + 16)

// The k-th smallest of the first ceil(k / L) keys of L sorted lists: an upper bound of the k-th smallest key overall (those k keys
// are all <= it) and a tight one - step 1 of vg_select_lists on its own.  Used by the filter kernels to take their start threshold
// straight from a pre-pass' per-CU lists (2 us in every workgroup) instead of waiting for a merge LAUNCH (11-15 us) to reduce them.
// Every thread of the workgroup calls (two barriers); scratch: (VG_SEL_MAX_HEADS + 1) * 8 bytes of LDS.  EMPTY: fewer than k keys.
#define VG_KTH_HEAD_SCRATCH_BYTES ((VG_SEL_MAX_HEADS + 1) * 8)
__device__ inline uint64_t vg_kth_head(const uint64_t *lists, int L, int k, uint8_t *scratch) {
    uint64_t *heads = reinterpret_cast<uint64_t *>(scratch);
    uint64_t *tau_slot = heads + VG_SEL_MAX_HEADS;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int m = (k + L - 1) / L;
    const int H = L * m;
    for (int h = tid; h < H; h += nthr) heads[h] = lists[(long long)(h / m) * VG_WAVE + (h % m)];
    if (tid == 0) *tau_slot = VG_EMPTY_KEY;
    __syncthreads();
    for (int h = tid; h < H; h += nthr) {
        const uint64_t mykey = heads[h];
        if (mykey == VG_EMPTY_KEY) continue;
        int rank = 0;
        for (int j = 0; j < H; ++j) rank += (heads[j] < mykey) ? 1 : 0;
        if (rank == k - 1) *tau_slot = mykey;
    }
    __syncthreads();
    return *tau_slot;
}

// lists : L lists of 64 slots each (ascending, VG_EMPTY_KEY padded; only slots < k are looked at); LDS or global
// out   : 64 slots, receives the k smallest ascending, VG_EMPTY_KEY padded
// scratch: VG_SEL_SCRATCH_BYTES of LDS, 16-byte aligned.  All threads of the workgroup must call (barriers inside).
// Requires L * ceil(k / L) <= VG_SEL_MAX_HEADS, i.e. L <= 256 and k <= 64.
__device__ inline void vg_select_lists(const uint64_t *lists, int L, int k, uint64_t *out, uint8_t *scratch) {
    uint64_t *heads = reinterpret_cast<uint64_t *>(scratch);
    uint64_t *surv = heads + VG_SEL_MAX_HEADS;
    unsigned int *nsurv = reinterpret_cast<unsigned int *>(surv + VG_SURV_CAP);
    uint64_t *tau_slot = reinterpret_cast<uint64_t *>(nsurv + 2);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int m = (k + L - 1) / L;
    const int H = L * m;

    for (int h = tid; h < H; h += nthr) heads[h] = lists[(long long)(h / m) * VG_WAVE + (h % m)];
    if (tid == 0) { *nsurv = 0; *tau_slot = VG_EMPTY_KEY; }
    __syncthreads();
    // rank of every head among the heads (unique keys; EMPTY heads tie with each other but never get rank k-1
    // unless fewer than k heads are valid, in which case tau stays EMPTY = "everything survives")
    for (int h = tid; h < H; h += nthr) {
        const uint64_t mykey = heads[h];
        if (mykey == VG_EMPTY_KEY) continue;
        int rank = 0;
        for (int j = 0; j < H; ++j) rank += (heads[j] < mykey) ? 1 : 0;
        if (rank == k - 1) *tau_slot = mykey;
    }
    __syncthreads();
    const uint64_t tau = *tau_slot;
    // survivors
    for (int s = tid; s < L * VG_WAVE; s += nthr) {
        if ((s & (VG_WAVE - 1)) >= k) continue;
        const uint64_t key = lists[s];
        if (key != VG_EMPTY_KEY && key <= tau) {
            const unsigned int pos = atomicAdd(nsurv, 1u);
            if (pos < VG_SURV_CAP) surv[pos] = key;
        }
    }
    for (int s = tid; s < VG_WAVE; s += nthr) out[s] = VG_EMPTY_KEY;
    __syncthreads();
    const int S = (int)min(*nsurv, (unsigned int)VG_SURV_CAP);
    for (int i = tid; i < S; i += nthr) {
        const uint64_t mykey = surv[i];
        int rank = 0;
        for (int j = 0; j < S; ++j) rank += (surv[j] < mykey) ? 1 : 0;
        if (rank < k) out[rank] = mykey;
    }
}
