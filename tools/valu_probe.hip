// valu_probe.hip - issue cost (cycles per wave64 instruction, one wave per SIMD, independent instruction stream) of the
// VALU ops the f16/bf16 scan kernels are made of.  Scratch measurement tool.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define PROBE(NAME, ASM, ...)                                                                             \
    __global__ void NAME(unsigned long long *out, float seed) {                                           \
        float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3;                                     \
        double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;                                    \
        unsigned u0 = (unsigned)seed;                                                                     \
        unsigned long long t0 = __builtin_readcyclecounter();                                             \
        for (int i = 0; i < 64; ++i) { asm volatile(REP64(ASM) : __VA_ARGS__); }                          \
        unsigned long long t1 = __builtin_readcyclecounter();                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                        \
        if (f0 + f1 + f2 + f3 + (float)(d0 + d1 + d2 + d3) + (float)u0 == 12345.f) out[1] = 1;             \
    }

PROBE(k_fma_f32, "v_fma_f32 %0, %1, %2, %0\n", "+v"(f0) : "v"(f1), "v"(f2))
PROBE(k_fma_f32_ind, "v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %1, %4, %5, %1\nv_fma_f32 %2, %4, %5, %2\nv_fma_f32 %3, %4, %5, %3\n", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(seed), "v"(seed))
PROBE(k_add_f64, "v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %4\nv_add_f64 %2, %2, %4\nv_add_f64 %3, %3, %4\n", "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)seed))
PROBE(k_fma_f64, "v_fma_f64 %0, %4, %4, %0\nv_fma_f64 %1, %4, %4, %1\nv_fma_f64 %2, %4, %4, %2\nv_fma_f64 %3, %4, %4, %3\n", "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)seed))
PROBE(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %4\nv_cvt_f64_f32 %3, %5\n", "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(f0), "v"(f1))
PROBE(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %4\nv_cvt_f32_f16 %1, %4\nv_cvt_f32_f16 %2, %4\nv_cvt_f32_f16 %3, %4\n", "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(u0))
PROBE(k_mul_f32, "v_mul_f32 %0, %4, %5\nv_mul_f32 %1, %4, %5\nv_mul_f32 %2, %4, %5\nv_mul_f32 %3, %4, %5\n", "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(seed), "v"(seed))
PROBE(k_pk_mul_f32, "v_pk_mul_f32 %0, %2, %2\nv_pk_mul_f32 %1, %2, %2\nv_pk_mul_f32 %0, %2, %2\nv_pk_mul_f32 %1, %2, %2\n", "=v"(d0), "=v"(d1) : "v"(d2))
PROBE(k_mul_f64, "v_mul_f64 %0, %4, %4\nv_mul_f64 %1, %4, %4\nv_mul_f64 %2, %4, %4\nv_mul_f64 %3, %4, %4\n", "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"((double)seed))
PROBE(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %4, %4, %0\nv_dot2_f32_f16 %1, %4, %4, %1\nv_dot2_f32_f16 %2, %4, %4, %2\nv_dot2_f32_f16 %3, %4, %4, %3\n", "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(u0))

PROBE(k_fma_mix_sub, "v_fma_mix_f32 %0, %4, 1.0, -%5 op_sel_hi:[1,0,1]\nv_fma_mix_f32 %1, %4, 1.0, -%5 op_sel:[1,0,1] op_sel_hi:[1,0,1]\nv_fma_mix_f32 %2, %5, 1.0, -%4 op_sel_hi:[1,0,1]\nv_fma_mix_f32 %3, %5, 1.0, -%4 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n", "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(u0), "v"(u0 + 7u))
PROBE(k_cvt_f32_f16_sdwa, "v_cvt_f32_f16_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n", "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(u0))
PROBE(k_sub_f32, "v_sub_f32 %0, %4, %5\nv_sub_f32 %1, %4, %5\nv_sub_f32 %2, %4, %5\nv_sub_f32 %3, %4, %5\n", "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(seed), "v"(seed))

template <typename K>
static void run(const char *name, K kern, int insts_per_rep, int waves) {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d, 1.5f);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d, 1.5f);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double n = 64.0 * 64.0 * insts_per_rep / (insts_per_rep == 1 ? 1 : 1);
    printf("%-16s waves/CU %2d : %.2f cycles per instruction per wave (s_memtime ticks)\n", name, waves, (double)h[0] / (64.0 * 64.0 * (insts_per_rep)));
    hipFree(d);
}

int main() {
    for (int waves : {4, 8, 16}) {          // 1, 2, 4 waves per SIMD
        run("v_fma_f32 dep", k_fma_f32, 1, waves);
        run("v_fma_f32 x4", k_fma_f32_ind, 4, waves);
        run("v_mul_f32", k_mul_f32, 4, waves);
        run("v_pk_mul_f32", k_pk_mul_f32, 4, waves);
        run("v_cvt_f32_f16", k_cvt_f32_f16, 4, waves);
        run("v_cvt_f64_f32", k_cvt_f64_f32, 4, waves);
        run("v_add_f64", k_add_f64, 4, waves);
        run("v_mul_f64", k_mul_f64, 4, waves);
        run("v_fma_f64", k_fma_f64, 4, waves);
        run("v_dot2_f32_f16", k_dot2_f32_f16, 4, waves);
        run("v_fma_mix_f32", k_fma_mix_sub, 4, waves);
        run("v_cvt_f32_f16 sdwa", k_cvt_f32_f16_sdwa, 4, waves);
        run("v_sub_f32", k_sub_f32, 4, waves);
    }
    return 0;
}
