// GPU-box tool: what ONE kernel launch + stream synchronize costs on this machine, independent of this repo's kernels - the floor
// under every single-query scan (DESIGN.md "per-query floor").  hipcc --offload-arch=gfx950 -O2 -o launch_floor_probe launch_floor_probe.hip
//   empty          an empty kernel, 1 workgroup
//   empty_x156     an empty kernel, 156 workgroups of 1024 threads (the grid of a 10k x 384 f32 scan)
//   host_rw        156 workgroups, each reads 1536 B of pinned host memory (the query) and writes 512 B to pinned host memory (its list)
//   two_launches   two empty kernels back to back, one synchronize (scan + merge)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ void k_host_rw(const float *q, unsigned long long *out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < 384; i += blockDim.x) s += q[i];
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (unsigned long long)(s != 12345.f) + threadIdx.x;
}

template <typename F>
static double p50_us(F f) {
    for (int i = 0; i < 200; ++i) f();
    std::vector<double> t;
    for (int i = 0; i < 2000; ++i) {
        auto a = std::chrono::steady_clock::now();
        f();
        t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float *hq; unsigned long long *hout;
    hipHostMalloc(&hq, 1536); hipHostMalloc(&hout, 256 * 512);
    for (int i = 0; i < 384; ++i) hq[i] = 1.f;
    printf("{\"empty_us\": %.2f", p50_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); hipStreamSynchronize(st); }));
    printf(", \"empty_x156_us\": %.2f", p50_us([&] { hipLaunchKernelGGL(k_empty, dim3(156), dim3(1024), 0, st); hipStreamSynchronize(st); }));
    printf(", \"host_rw_us\": %.2f", p50_us([&] { hipLaunchKernelGGL(k_host_rw, dim3(156), dim3(1024), 0, st, hq, hout); hipStreamSynchronize(st); }));
    printf(", \"two_launches_us\": %.2f}\n", p50_us([&] { hipLaunchKernelGGL(k_empty, dim3(156), dim3(1024), 0, st); hipLaunchKernelGGL(k_empty, dim3(1), dim3(1024), 0, st); hipStreamSynchronize(st); }));
    return 0;
}
