// read_probe.hip - pure HBM read ceiling of the device: the number the scan kernel's achieved GB/s should be judged
// against next to the 8 TB/s spec figure (SURVEY 8d "Peaks").  Scratch measurement tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/read_probe.hip -o gpurun_out/read_probe && gpurun_out/read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(1024) void read_kernel(const u32x4 *p, size_t n16, unsigned *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;      // never true for the fill pattern; keeps the loads
}

template <int U, bool NT>
static void run(const u32x4 *d, size_t n16, unsigned *sink, int blocks, const char *tag) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(blocks), dim3(1024), 0, 0, d, n16, sink);
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((read_kernel<U, NT>), dim3(blocks), dim3(1024), 0, 0, d, n16, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best; sum += ms;
    }
    const double gb = (double)n16 * 16 / 1e9;
    printf("%-22s blocks %5d : mean %.3f ms %.0f GB/s   best %.3f ms %.0f GB/s\n", tag, blocks, sum / reps, gb / (sum / reps) * 1e3, best, gb / best * 1e3);
}

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? (size_t)atof(argv[1]) : 15.36e9);
    const size_t n16 = bytes / 16;
    u32x4 *d; unsigned *sink;
    if (hipMalloc(&d, n16 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4);
    hipMemset(d, 0x5A, n16 * 16);
    hipDeviceSynchronize();
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) == hipSuccess) {
        printf("device: %s (%s)  CUs %d  core clock %.0f MHz  memory clock %.0f MHz  bus %d bit  L2 %.1f MiB  HBM %.1f GiB\n", pr.name,
               pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1e3, pr.memoryClockRate / 1e3, pr.memoryBusWidth,
               pr.l2CacheSize / 1048576.0, pr.totalGlobalMem / 1073741824.0);
        printf("peak from these: %.2f TB/s (bus / 8 x reported memory clock x 4 transfers per clock = 8 Gb/s per pin)\n",
               pr.memoryBusWidth / 8.0 * pr.memoryClockRate * 1e3 * 4 / 1e12);
    }
    printf("pure read of %.2f GB\n", n16 * 16 / 1e9);
    for (int blocks : {256, 512, 1024}) {
        run<4, true>(d, n16, sink, blocks, "U=4  nontemporal");
        run<8, true>(d, n16, sink, blocks, "U=8  nontemporal");
        run<12, true>(d, n16, sink, blocks, "U=12 nontemporal");
        run<8, false>(d, n16, sink, blocks, "U=8  plain");
    }
    // device-to-device copy ceiling for comparison (read + write)
    u32x4 *d2;
    if (hipMalloc(&d2, n16 * 16) == hipSuccess) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipMemcpy(d2, d, n16 * 16, hipMemcpyDeviceToDevice);
        hipEventRecord(a, 0);
        for (int r = 0; r < 5; ++r) hipMemcpyAsync(d2, d, n16 * 16, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("hipMemcpy D2D: %.3f ms per copy, %.0f GB/s moved (read+write)\n", ms / 5, 2.0 * n16 * 16 / 1e9 / (ms / 5) * 1e3);
    }
    return 0;
}
