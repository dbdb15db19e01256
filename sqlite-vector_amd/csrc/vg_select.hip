// vg_select.hip - top-k for k > 64 (beyond the fused one-slot-per-lane list): everything stays on the device.
//
//   1. the scan kernel runs in store mode and writes all N clamped distances;
//   2. vg_keys_kernel packs (order-preserving distance image << 32 | position), VG_EMPTY_KEY for NaN / +Inf rows
//      (they never enter the reference's slots: sqlite-vector.c:1809, :2102);
//   3. one device radix sort of the N 64-bit keys (rocPRIM, ascending) - keys are unique, so the first k entries
//      ARE the answer in the contract's (distance, scan position) order;
//   4. the host copies k keys back and decodes them.
//
// This is the rare path (the reference's typical k is 10-100); its cost is a few N-sized passes on the device.
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
