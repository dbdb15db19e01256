// vg_select.hip - top-k for k > 64 (beyond the fused one-slot-per-lane list): everything stays on the device.
//
//   1. the scan kernel runs in store mode and writes all N clamped distances;
//   2. vg_keys_kernel packs (order-preserving distance image << 32 | position), VG_EMPTY_KEY for NaN / +Inf rows
//      (they never enter the reference's slots: sqlite-vector.c:1809, :2102);
//   3. one device radix sort of the N 64-bit keys (rocPRIM, ascending) - keys are unique, so the first k entries
//      ARE the answer in the contract's (distance, scan position) order;
//   4. the host copies k keys back and decodes them.
//
// That full sort costs ~0.85 ms at N = 10M for a k of 100.  vg_select_topk_keys below replaces it with a RADIX SELECT:
// three histogram passes over the 32-bit distance images (11 + 11 + 10 bits, each pass only counting the elements
// that match the prefix found so far) pin the image T of the k-th smallest distance; one gather pass collects the
// keys of every element with image <= T (k plus whatever ties T has); only those are sorted.  The full sort stays as
// the fallback for degenerate inputs (so many ties at T that the gathered set is a large part of N).
// A separate translation unit keeps rocPRIM's templates out of the scan kernels' compile.
#include <cstring>               // rocPRIM's texture_cache_iterator.hpp calls the host memset without including it
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdint.h>

#include "vg_device.h"

__global__ void vg_keys_kernel(const float *dist, long long n, uint64_t *keys) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float d = dist[i];
        keys[i] = (d < INFINITY) ? vg_make_key(d, (uint32_t)i) : VG_EMPTY_KEY;
    }
}

// temp storage query (bytes) for sorting n keys
extern "C" int vg_select_temp_bytes(long long n, size_t *bytes) {
    uint64_t *nullk = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, b, nullk, nullk, (size_t)n, 0, 64, (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    *bytes = b;
    return 0;
}

// dist[n] -> keys_tmp[n] -> keys_sorted[n] (ascending).  All buffers are device memory; asynchronous on `stream`.
extern "C" int vg_select_sorted_keys(const float *dist, long long n, uint64_t *keys_tmp, uint64_t *keys_sorted,
                                     void *temp, size_t temp_bytes, hipStream_t stream) {
    const int threads = 256;
    long long blocks = (n + threads - 1) / threads;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vg_keys_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, dist, n, keys_tmp);
    hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, keys_tmp, keys_sorted, (size_t)n, 0, 64, stream, false);
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------------ radix select

struct VgSelState {
    uint32_t prefix;      // image bits decided so far (high bits), low bits zero
    uint32_t k_rem;       // rank still wanted inside the current prefix class (1-based)
    uint32_t gathered;    // number of keys written by the gather pass
    uint32_t finite;      // elements with a finite-or--Inf distance (image != 0xFFFFFFFF ... i.e. d < +Inf)
};

#define VG_SEL_BINS 2048

__device__ inline uint32_t vg_dist_image(float d) { return (d < INFINITY) ? vg_f32_sortable(d) : 0xFFFFFFFFu; }

// counts, per bin of `bits` bits at `shift`, the elements whose image matches st->prefix on the bits above them.
// Distances of one query share their exponent, so the first pass sends almost every element to one or two bins:
// each thread run-length-compresses its own stream (same bin as the previous element -> a register increment) and
// only a change of bin costs an LDS atomic; the later passes spread over all bins and do not contend.
__global__ __launch_bounds__(1024) void vg_sel_hist_kernel(const float *dist, long long n, int shift, int bits, const VgSelState *st,
                                                           uint32_t *hist) {
    __shared__ uint32_t h[VG_SEL_BINS];
    for (int i = threadIdx.x; i < VG_SEL_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint32_t prefix = st->prefix;
    const uint32_t above = (shift + bits >= 32) ? 0u : (0xFFFFFFFFu << (shift + bits));
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t run_bin = 0xFFFFFFFFu, run_cnt = 0;
    auto count = [&](float d) {
        const uint32_t img = vg_dist_image(d);
        if (img != 0xFFFFFFFFu && (img & above) == prefix) {
            const uint32_t b = (img >> shift) & mask;
            if (b == run_bin) { ++run_cnt; }
            else {
                if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
                run_bin = b; run_cnt = 1;
            }
        }
    };
    const long long n4 = n >> 2;
    const float4 *d4 = reinterpret_cast<const float4 *>(dist);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = d4[i];
        count(v.x); count(v.y); count(v.z); count(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) count(dist[(n4 << 2) + threadIdx.x]);
    if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
    __syncthreads();
    for (int b = threadIdx.x; b < VG_SEL_BINS; b += blockDim.x)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}

// one workgroup: the bin holding rank k_rem; narrows the prefix, re-bases k_rem, clears the histogram for the next pass
__global__ __launch_bounds__(1024) void vg_sel_pick_kernel(uint32_t *hist, int shift, int bits, int first_pass, uint32_t k, VgSelState *st) {
    __shared__ uint32_t part[1024];          // inclusive scan of per-thread sums (two bins per thread)
    __shared__ uint32_t found_bin, found_below;
    const int nb = 1 << bits, t = threadIdx.x;
    const uint32_t c0 = (2 * t < nb) ? hist[2 * t] : 0u, c1 = (2 * t + 1 < nb) ? hist[2 * t + 1] : 0u;
    part[t] = c0 + c1;
    if (t == 0) { found_bin = (uint32_t)(nb - 1); found_below = 0u; }
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                    // Hillis-Steele inclusive scan
        const uint32_t v = (t >= off) ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t total = part[1023];
    uint32_t want = first_pass ? k : st->k_rem;
    if (want > total) want = total;                               // fewer than k finite rows: everything finite qualifies
    const uint32_t below = part[t] - (c0 + c1);                   // elements in bins before 2t
    if (want > 0) {
        if (c0 > 0 && below < want && want <= below + c0) { found_bin = 2 * t; found_below = below; }
        else if (c1 > 0 && below + c0 < want && want <= below + c0 + c1) { found_bin = 2 * t + 1; found_below = below + c0; }
    }
    __syncthreads();
    if (t == 0) {
        if (first_pass) st->finite = total;
        st->prefix = (first_pass ? 0u : st->prefix) | (found_bin << shift);
        st->k_rem = want - found_below;
    }
    hist[2 * t] = 0;
    hist[2 * t + 1] = 0;
}

// every element with image <= T (T = st->prefix after the third pass), as packed keys, in arbitrary order
__global__ __launch_bounds__(256) void vg_sel_gather_kernel(const float *dist, long long n, VgSelState *st, uint64_t *out, uint32_t cap) {
    const uint32_t T = st->prefix;
    auto take = [&](float d, long long i) {
        const uint32_t img = vg_dist_image(d);
        if (img != 0xFFFFFFFFu && img <= T) {
            const uint32_t slot = atomicAdd(&st->gathered, 1u);
            if (slot < cap) out[slot] = ((uint64_t)img << 32) | (uint32_t)i;
        }
    };
    const long long n4 = n >> 2;
    const float4 *d4 = reinterpret_cast<const float4 *>(dist);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = d4[i];
        take(v.x, 4 * i); take(v.y, 4 * i + 1); take(v.z, 4 * i + 2); take(v.w, 4 * i + 3);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) take(dist[(n4 << 2) + threadIdx.x], (n4 << 2) + threadIdx.x);
}

// dist[n] -> the k smallest keys, ascending, in keys_sorted[0..k) (fewer when fewer rows are finite: *out_count).
// state: 4 words + VG_SEL_BINS words of device scratch.  keys_tmp / keys_sorted hold `cap` keys.  Synchronises the
// stream once (the gathered count decides the size of the final sort).  Returns 1 when the gathered set did not fit
// `cap` (caller falls back to vg_select_sorted_keys), 0 on success, a hipError_t otherwise.
extern "C" int vg_select_topk_keys(const float *dist, long long n, uint32_t k, uint64_t *keys_tmp, uint64_t *keys_sorted,
                                   uint32_t cap, void *temp, size_t temp_bytes, uint32_t *state, hipStream_t stream,
                                   uint32_t *out_count) {
    VgSelState *st = reinterpret_cast<VgSelState *>(state);
    uint32_t *hist = state + 4;
    hipError_t e = hipMemsetAsync(state, 0, (4 + VG_SEL_BINS) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    long long blocks = (n / 4 + 1023) / 1024;
    if (blocks > 256 * 2) blocks = 256 * 2;
    if (blocks < 1) blocks = 1;
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(vg_sel_hist_kernel, dim3((unsigned)blocks), dim3(1024), 0, stream, dist, n, shifts[p], bits[p], st, hist);
        hipLaunchKernelGGL(vg_sel_pick_kernel, dim3(1), dim3(1024), 0, stream, hist, shifts[p], bits[p], p == 0 ? 1 : 0, k, st);
    }
    long long gblocks = (n / 4 + 255) / 256;
    if (gblocks > 256 * 16) gblocks = 256 * 16;
    if (gblocks < 1) gblocks = 1;
    hipLaunchKernelGGL(vg_sel_gather_kernel, dim3((unsigned)gblocks), dim3(256), 0, stream, dist, n, st, keys_tmp, cap);
    VgSelState h;
    e = hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return (int)e;
    if (h.gathered > cap) return 1;
    if (h.gathered > 0) {
        size_t tb = temp_bytes;
        e = rocprim::radix_sort_keys(temp, tb, keys_tmp, keys_sorted, (size_t)h.gathered, 0, 64, stream, false);
        if (e != hipSuccess) return (int)e;
    }
    *out_count = h.gathered < k ? h.gathered : k;
    return (int)hipGetLastError();
}
