// GPU-box tool: issue rate of the two int8 MFMA shapes on gfx950 - independent accumulator chains against ONE dependent chain, one / two
// wavefronts per SIMD (what vg_batch_q8.hip's two query sets per wavefront were built on).  hipcc --offload-arch=gfx950 -O3 -o mfma_i8_probe tools/mfma_i8_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE, int CHAINS>
__global__ void probe(int iters, int *out, unsigned long long *cycles) {
    i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    i32x16 c32[CHAINS]; i32x4 c16[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { for (int r = 0; r < 16; ++r) c32[i][r] = 0; c16[i] = i32x4{0, 0, 0, 0}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if constexpr (SHAPE == 32) c32[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c32[i], 0, 0, 0);
            else c16[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c16[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < CHAINS; ++i) { for (int r = 0; r < 16; ++r) s += c32[i][r]; s += c16[i][0] + c16[i][1] + c16[i][2] + c16[i][3]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
template <int SHAPE, int CHAINS>
void run(const char *name, int threads) {
    int *out; unsigned long long *cyc, h = 0;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<SHAPE, CHAINS><<<256, threads>>>(100, out, cyc);
    hipEventRecord(e0);
    probe<SHAPE, CHAINS><<<256, threads>>>(iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * CHAINS;
    const double ops = 2.0 * (SHAPE == 32 ? 32.0 * 32 * 32 : 16.0 * 16 * 64) * n * 256 * (threads / 64);
    printf("%-28s waves/SIMD %d  %.1f cycles per MFMA per wave  (%.3f ms, %.0f TOP/s chip)\n", name, threads / 256, (double)h / n, ms, ops / ms / 1e9);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<32, 4>("32x32x32 i8, 4 chains", 256); run<32, 4>("32x32x32 i8, 4 chains", 512);
    run<16, 8>("16x16x64 i8, 8 chains", 256); run<16, 8>("16x16x64 i8, 8 chains", 512);
    run<16, 2>("16x16x64 i8, 2 chains", 256); run<32, 1>("32x32x32 i8, 1 chain", 256);
    return 0;
}
