// scratch probe: where do the 24 us of vg_merge_kernel go?  build: hipcc --offload-arch=gfx950 -O3 -I../sqlite-vector_amd/csrc merge_probe.hip -o merge_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "vg_scan.h"

__global__ __launch_bounds__(1024) void k_empty(const uint64_t *cand, int nlists, int k, uint64_t *out) {
    if (threadIdx.x == 0 && nlists < 0) out[0] = cand[0];
}
__global__ __launch_bounds__(1024) void k_loads(const uint64_t *cand, int nlists, int k, uint64_t *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { int l = wave + j * 16; if (l < nlists) acc ^= cand[(long long)l * 64 + lane]; }
    if (acc == 0x1234567) out[lane] = acc;
}
__global__ __launch_bounds__(256) void k_loads256(const uint64_t *cand, int nlists, int k, uint64_t *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t acc = 0;
    for (int l = wave; l < nlists; l += 4) acc ^= cand[(long long)l * 64 + lane];
    if (acc == 0x1234567) out[lane] = acc;
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}
int main() {
    const int nlists = 256, k = 20;
    std::vector<uint64_t> h((size_t)nlists * 64, VG_EMPTY_KEY);
    for (int l = 0; l < nlists; ++l) { std::vector<uint64_t> v(k); for (int i = 0; i < k; ++i) v[i] = ((uint64_t)(0x80000000u + (uint32_t)rand()) << 32) | (uint32_t)(l * 1000 + i); std::sort(v.begin(), v.end()); for (int i = 0; i < k; ++i) h[(size_t)l * 64 + i] = v[i]; }
    uint64_t *d, *o; hipMalloc(&d, h.size() * 8); hipMalloc(&o, 64 * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    printf("empty1024   %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(1024), 0, 0, d, nlists, k, o); }, 200));
    printf("loads1024   %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(k_loads, dim3(1), dim3(1024), 0, 0, d, nlists, k, o); }, 200));
    printf("loads256    %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(k_loads256, dim3(1), dim3(256), 0, 0, d, nlists, k, o); }, 200));
    printf("merge n=16  %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(vg_merge_kernel, dim3(1), dim3(1024), 0, 0, d, 16, k, o); }, 200));
    printf("merge n=256 %.2f us/launch\n", timeit([&] { hipLaunchKernelGGL(vg_merge_kernel, dim3(1), dim3(1024), 0, 0, d, nlists, k, o); }, 200));
    std::vector<uint64_t> r(64); hipMemcpy(r.data(), o, 64 * 8, hipMemcpyDeviceToHost);
    std::vector<uint64_t> all; for (auto x : h) if (x != VG_EMPTY_KEY) all.push_back(x); std::sort(all.begin(), all.end());
    bool ok = true; for (int i = 0; i < k; ++i) ok &= (r[i] == all[i]);
    printf("merge result %s\n", ok ? "OK" : "WRONG");
    return 0;
}
