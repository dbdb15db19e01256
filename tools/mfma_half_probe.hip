// mfma_half_probe.hip - what v_mfma_f32_32x32x16_{f16,bf16} does with subnormal inputs and how it rounds its sums.
// The half-precision batch kernel (vg_batch_h.hip) uses these instructions as a FILTER whose error bound has to hold.
//     hipcc --offload-arch=gfx950 -O2 -o mfma_half_probe tools/mfma_half_probe.hip && ./mfma_half_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cmath>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A[i][k], B[k][j]: lane (x, h) holds k = 8h .. 8h+7 of row/column x.  Every row i of A = avals, every column j of B = bvals.
__global__ void probe_f16(const uint16_t *avals, const uint16_t *bvals, float cin, float *out) {
    const int h = threadIdx.x >> 5;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = __builtin_bit_cast(_Float16, avals[8 * h + j]);
        b[j] = __builtin_bit_cast(_Float16, bvals[8 * h + j]);
    }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = cin;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void probe_bf16(const uint16_t *avals, const uint16_t *bvals, float cin, float *out) {
    const int h = threadIdx.x >> 5;
    b8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = __builtin_bit_cast(__bf16, avals[8 * h + j]);
        b[j] = __builtin_bit_cast(__bf16, bvals[8 * h + j]);
    }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = cin;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

static float run(bool bf, const float *a, const float *b, float cin) {
    uint16_t ha[16], hb[16], *da, *db;
    float *dout, out = 0;
    for (int i = 0; i < 16; ++i) { ha[i] = bf ? f2b(a[i]) : f2h(a[i]); hb[i] = bf ? f2b(b[i]) : f2h(b[i]); }
    hipMalloc(&da, 32); hipMalloc(&db, 32); hipMalloc(&dout, 4);
    hipMemcpy(da, ha, 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, 32, hipMemcpyHostToDevice);
    if (bf) hipLaunchKernelGGL(probe_bf16, dim3(1), dim3(64), 0, 0, da, db, cin, dout);
    else hipLaunchKernelGGL(probe_f16, dim3(1), dim3(64), 0, 0, da, db, cin, dout);
    hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return out;
}

int main() {
    for (int bf = 0; bf < 2; ++bf) {
        const char *nm = bf ? "bf16" : "f16";
        float a[16] = {0}, b[16] = {0};
        // 1. subnormal input: f16 2^-20 (subnormal) x 1024 = 2^-10 ; bf16 2^-130 x 2^100 = 2^-30
        a[0] = bf ? ldexpf(1.0f, -130) : ldexpf(1.0f, -20); b[0] = bf ? ldexpf(1.0f, 100) : 1024.0f;
        printf("%s subnormal A input   : got %.9g want %.9g\n", nm, run(bf, a, b, 0.0f), bf ? ldexp(1.0, -30) : ldexp(1.0, -10));
        printf("%s subnormal B input   : got %.9g want %.9g\n", nm, run(bf, b, a, 0.0f), bf ? ldexp(1.0, -30) : ldexp(1.0, -10));
        // 2. subnormal x subnormal-ish product far below f16 range but normal in f32: 2^-20 * 2^-14
        if (!bf) { a[0] = ldexpf(1.0f, -20); b[0] = ldexpf(1.0f, -14); printf("%s tiny product        : got %.9g want %.9g\n", nm, run(bf, a, b, 0.0f), ldexp(1.0, -34)); }
        // 3. rounding of the sum: C = 1, fifteen products of 1.5 * 2^-24 each (0.75 ulp of 1)
        for (int i = 0; i < 16; ++i) { a[i] = (i < 15) ? 1.5f : 0.0f; b[i] = (i < 15) ? ldexpf(1.0f, bf ? -24 : -24) : 0.0f; }
        if (!bf) for (int i = 0; i < 15; ++i) { a[i] = 1.5f * ldexpf(1.0f, -12); b[i] = ldexpf(1.0f, -12); }
        const float got = run(bf, a, b, 1.0f);
        printf("%s sum rounding        : got 1 + %.3f ulp  (exact sum 11.25 ulp; sequential RNE 15; truncation 0..11)\n", nm, (got - 1.0f) / ldexpf(1.0f, -23));
        // 4. cancellation inside one instruction: 1e4*1e4 - 1e4*1e4 + 1 (products exact in f32)
        for (int i = 0; i < 16; ++i) { a[i] = 0; b[i] = 0; }
        a[0] = 4096; b[0] = 4096; a[1] = -4096; b[1] = 4096; a[2] = 1; b[2] = ldexpf(1.0f, -10);
        printf("%s in-instruction cancel: got %.9g want %.9g\n", nm, run(bf, a, b, 0.0f), ldexp(1.0, -10));
        a[1] = 0; a[9] = -4096; b[9] = 4096;          // the cancelling product in the other half-wave's k range
        printf("%s cross-half cancel   : got %.9g want %.9g\n", nm, run(bf, a, b, 0.0f), ldexp(1.0, -10));
    }
    return 0;
}
