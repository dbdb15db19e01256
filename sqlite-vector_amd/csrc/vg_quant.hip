// vg_quant.hip - vector_quantize on the GPU: the two passes of vector_rebuild_quantization
// (sqlite-vector.c:1147-1336) over the corpus that is already staged in HBM.
//
//   pass 1  vg_minmax_kernel    global min / max / any-negative over every element (as float, the way the
//                               reference widens each type, :1228-1254); NaN elements never win a comparison
//   pass 2  vg_quantize_kernel  s = (v - offset) * scale, round half away from zero, clamp to u8 / i8 - the same
//                               float operations, one rounding each, as quantize_* (:495-757), including the
//                               unguarded float->int conversion of the f32 source path (x86 cvttss2si: NaN / out of
//                               range -> INT_MIN) and the NaN/Inf rules of q_round_u8/s8 for the other types.
// Output of pass 2 is a tightly packed N x dim byte matrix; the extension host interleaves the rowids and writes the
// reference's persisted record format.  Bit-exactness is tested against the pinned oracle quantizer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "vg_device.h"
#include "vg_half.h"

// Both passes share one decomposition (the scan kernels' own): 16 lanes per row, lane l owns the 16-byte chunks l, l + 16, ... of
// the row - every load is a full 16 bytes of a 256-byte run, no index division anywhere (round 2's first draft did a 64-bit
// divide + modulo per ELEMENT).  A lane first issues the loads of G of its chunks - unconditionally, a chunk behind the row's
// last one reads 16 zero bytes instead (a load under a branch is waited for on the spot: the first form of these kernels had ONE
// load in flight per wavefront and ran at 0.59 / 0.73 of the HBM peak) - and then works through them.  G = chunks per lane of the
// row, up to 8 (rows of up to 128 chunks in one go, longer ones in rounds of 128).  The zero padding behind `dim` elements is
// masked out: it must not become a minimum.
typedef uint32_t vgq_u32x4 __attribute__((ext_vector_type(4)));
static __device__ __attribute__((aligned(16))) uint32_t vgq_zero_chunk[4] = {0u, 0u, 0u, 0u};
__device__ inline uint4 vgq_load16(const uint8_t *p) {      // every corpus byte is read once per pass: keep it out of the caches
    const vgq_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const vgq_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int G>
__device__ inline void vgq_load_part(uint4 (&buf)[G], const uint8_t *row, int c0, int l16, int nch) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = c0 + g * 16 + l16;
        buf[g] = vgq_load16(c < nch ? row + (long long)c * 16 : reinterpret_cast<const uint8_t *>(vgq_zero_chunk));
    }
}
template <int VT> struct VgqChunk {                         // elements of one 16-byte chunk, widened the way the reference does
    static constexpr int N = (VT == T_F32) ? 4 : (VT == T_F16 || VT == T_BF16) ? 8 : 16;
    __device__ static inline void widen(const uint4 &v, float (&out)[N]) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if constexpr (VT == T_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = __uint_as_float(w[j]);
        } else if constexpr (VT == T_F16 || VT == T_BF16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (VT == T_F16) { out[2 * j] = vg_h2f((uint16_t)(w[j] & 0xFFFFu)); out[2 * j + 1] = vg_h2f((uint16_t)(w[j] >> 16)); }
                else { out[2 * j] = vg_b2f((uint16_t)(w[j] & 0xFFFFu)); out[2 * j + 1] = vg_b2f((uint16_t)(w[j] >> 16)); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t byte = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                out[j] = (VT == T_U8) ? (float)byte : (float)(int)(int8_t)byte;
            }
        }
    }
};

// out[0] = sortable(min), out[1] = sortable(max), out[2] = any negative.  Pre-set by the host to
// sortable(FLT_MAX), sortable(-FLT_MAX), 0 (the reference's initial values, :1197-1198).
template <int VT, int G>
__global__ __launch_bounds__(256) void vg_minmax_kernel(const uint8_t *rows, long long n_rows, long long stride, int dim, int nch,
                                                         uint32_t *out) {
    constexpr int N = VgqChunk<VT>::N;
    const int l16 = threadIdx.x & 15;
    const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ngroups = ((long long)gridDim.x * blockDim.x) >> 4;
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;     // ("any negative element" is min < 0: NaN never wins a comparison, -0 is not < 0)
    const int full = dim / N;                                // chunks without padding
    for (long long r = group; r < n_rows; r += ngroups) {
        const uint8_t *p = rows + r * stride;
        for (int c0 = 0; c0 < nch; c0 += 16 * G) {
            uint4 buf[G];
            vgq_load_part<G>(buf, p, c0, l16, nch);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int c = c0 + g * 16 + l16;
                float e[N];
                VgqChunk<VT>::widen(buf[g], e);
                if (c < full) {                                               // a chunk of elements only
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        if (e[j] < lo) lo = e[j];
                        if (e[j] > hi) hi = e[j];
                    }
                } else {                                                      // the row's last chunk (or nothing: live <= 0)
                    const int live = dim - c * N;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const bool in = j < live;
                        if (in && e[j] < lo) lo = e[j];
                        if (in && e[j] > hi) hi = e[j];
                    }
                }
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        if (l2 < lo) lo = l2;
        if (h2 > hi) hi = h2;
    }
    // one set of atomics per WORKGROUP, and only where it can change the result: the first form sent three same-address atomics per
    // wavefront (98k of them over a 4M-row corpus, serialized in one L2 channel - a third of the kernel's time)
    __shared__ float wlo[4], whi[4];
    if ((threadIdx.x & 63) == 0) { wlo[threadIdx.x >> 6] = lo; whi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            if (wlo[w] < lo) lo = wlo[w];
            if (whi[w] > hi) hi = whi[w];
        }
        const uint32_t klo = vg_f32_sortable(lo), khi = vg_f32_sortable(hi);
        if (klo < __hip_atomic_load(&out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&out[0], klo);
        if (khi > __hip_atomic_load(&out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&out[1], khi);
        if (lo < 0.0f && !__hip_atomic_load(&out[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&out[2], 1u);
    }
}

__device__ inline int vgq_trunc_x86(float r) {           // cvttss2si semantics
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) return (int)0x80000000u;
    return (int)r;
}

// one element: s = (v - offset) * scale (two roundings, no contraction), round half away from zero, clamp
template <int VT>
__device__ inline uint32_t vgq_one(float v, float scale, float offset, int qtype_u8) {
    const float s = (v - offset) * scale;
    const float rr = s + 0.5f * (1.0f - 2.0f * (s < 0.0f ? 1.0f : 0.0f));
    if constexpr (VT == T_F32) {                                          // sqlite-vector.c:517-548, :626-656
        const int ir = vgq_trunc_x86(rr);
        if (qtype_u8) return (uint32_t)(ir > 255 ? 255 : (ir < 0 ? 0 : ir));
        return (uint32_t)(uint8_t)(int8_t)(ir > 127 ? 127 : (ir < -128 ? -128 : ir));
    } else if (qtype_u8) {                                                // q_round_u8, :495-504
        if (!isfinite(s)) return (s > 0.0f) ? 255u : 0u;
        if (rr >= 255.0f) return 255u;
        if (rr <= 0.0f) return 0u;
        return (uint32_t)(uint8_t)(int)rr;
    } else {                                                              // q_round_s8, :506-515
        int8_t t;
        if (!isfinite(s)) t = (s > 0.0f) ? 127 : (s < 0.0f ? -128 : 0);
        else if (rr >= 127.0f) t = 127;
        else if (rr <= -128.0f) t = -128;
        else t = (int8_t)(int)rr;
        return (uint32_t)(uint8_t)t;
    }
}

// rows [row0, row0 + n_rows) -> n_rows x dim tightly packed bytes.  A chunk's N output bytes go out as one 4- / 8- / 16-byte
// store when the packed row length keeps them aligned (dim a multiple of N), byte by byte otherwise.
template <int VT, int G>
__global__ __launch_bounds__(256) void vg_quantize_kernel(const uint8_t *rows, long long row0, long long n_rows, long long stride,
                                                           int dim, int nch, float scale, float offset, int qtype_u8, uint8_t *out) {
    constexpr int N = VgqChunk<VT>::N;
    const int l16 = threadIdx.x & 15;
    const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ngroups = ((long long)gridDim.x * blockDim.x) >> 4;
    const bool packed_ok = (dim % N) == 0;
    const int full = dim / N;
    for (long long r = group; r < n_rows; r += ngroups) {
        const uint8_t *p = rows + (row0 + r) * stride;
        uint8_t *o = out + r * (long long)dim;
        for (int c0 = 0; c0 < nch; c0 += 16 * G) {
            uint4 buf[G];
            vgq_load_part<G>(buf, p, c0, l16, nch);
            if constexpr (N == 4 && G >= 4) {
                // f32 rows: a chunk becomes ONE output dword, and 64 lanes x 4 bytes per store instruction left the pass store-issue bound
                // (0.75-0.80 of the HBM peak).  Four chunk groups at a time, the four lanes of a quad transpose their 4 x 4 dwords (two
                // DPP exchange stages) so that lane i holds the 16 CONTIGUOUS bytes of group i: one dwordx4 store per lane and four groups.
                if (packed_ok && (dim & 15) == 0 && c0 + 64 <= full) {
                    uint32_t a[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float e[N];
                        VgqChunk<VT>::widen(buf[g], e);
                        a[g] = vgq_one<VT>(e[0], scale, offset, qtype_u8) | (vgq_one<VT>(e[1], scale, offset, qtype_u8) << 8) |
                               (vgq_one<VT>(e[2], scale, offset, qtype_u8) << 16) | (vgq_one<VT>(e[3], scale, offset, qtype_u8) << 24);
                    }
                    const bool odd = (l16 & 1) != 0, hi = (l16 & 2) != 0;
                    {   // lanes i, i ^ 1 swap what the other one needs of (a0, a1) and of (a2, a3)
                        const uint32_t r01 = vg_dpp_u32<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(odd ? a[0] : a[1]);
                        const uint32_t r23 = vg_dpp_u32<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(odd ? a[2] : a[3]);
                        if (odd) { a[0] = r01; a[2] = r23; } else { a[1] = r01; a[3] = r23; }
                    }
                    {   // lanes i, i ^ 2 swap (a0, a1) against (a2, a3)
                        const uint32_t r0 = vg_dpp_u32<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(hi ? a[0] : a[2]);
                        const uint32_t r1 = vg_dpp_u32<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(hi ? a[1] : a[3]);
                        if (hi) { a[0] = r0; a[1] = r1; } else { a[2] = r0; a[3] = r1; }
                    }
                    // lane i of the quad now holds the dwords of its quad's four lanes for group i (in lane order: a0 from lane 0 ...)
                    const int cq = c0 + (l16 & 3) * 16 + (l16 & ~3);
                    *reinterpret_cast<uint4 *>(o + cq * 4) = make_uint4(a[0], a[1], a[2], a[3]);
#pragma unroll
                    for (int g = 4; g < G; ++g) {
                        const int c = c0 + g * 16 + l16;
                        float e[N];
                        VgqChunk<VT>::widen(buf[g], e);
                        if (c < full)
                            *reinterpret_cast<uint32_t *>(o + c * 4) = vgq_one<VT>(e[0], scale, offset, qtype_u8) | (vgq_one<VT>(e[1], scale, offset, qtype_u8) << 8) |
                                                                       (vgq_one<VT>(e[2], scale, offset, qtype_u8) << 16) | (vgq_one<VT>(e[3], scale, offset, qtype_u8) << 24);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int c = c0 + g * 16 + l16;
                float e[N];
                VgqChunk<VT>::widen(buf[g], e);
                uint32_t q[N];
#pragma unroll
                for (int j = 0; j < N; ++j) q[j] = vgq_one<VT>(e[j], scale, offset, qtype_u8);
                if (packed_ok && c < full) {
                    uint32_t w[N / 4];
#pragma unroll
                    for (int j = 0; j < N / 4; ++j) w[j] = q[4 * j] | (q[4 * j + 1] << 8) | (q[4 * j + 2] << 16) | (q[4 * j + 3] << 24);
                    if constexpr (N == 4) *reinterpret_cast<uint32_t *>(o + c * 4) = w[0];
                    else if constexpr (N == 8) *reinterpret_cast<uint2 *>(o + c * 8) = make_uint2(w[0], w[1]);
                    else *reinterpret_cast<uint4 *>(o + c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
                } else if (c * N < dim) {
#pragma unroll
                    for (int j = 0; j < N; ++j) if (c * N + j < dim) o[c * N + j] = (uint8_t)q[j];
                }
            }
        }
    }
}

static inline unsigned vgq_blocks(long long n_rows) {
    long long blocks = (n_rows * 16 + 255) / 256;            // 16 lanes per row
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// chunks per lane of one row (16 lanes per row), rounded up to an instantiated G; rows beyond 128 chunks go round by round at 8
static inline int vgq_pick_g(int nch) {
    const int need = (nch + 15) / 16;
    static const int gs[] = {1, 2, 3, 4, 6, 8};
    for (int g : gs) if (g >= need) return g;
    return 8;
}
template <int VT> static void vgq_minmax_go(int G, dim3 g, dim3 b, hipStream_t stream, const uint8_t *rows, long long n_rows, long long stride,
                                            int dim, int nch, uint32_t *out) {
    switch (G) {
        case 1: hipLaunchKernelGGL((vg_minmax_kernel<VT, 1>), g, b, 0, stream, rows, n_rows, stride, dim, nch, out); break;
        case 2: hipLaunchKernelGGL((vg_minmax_kernel<VT, 2>), g, b, 0, stream, rows, n_rows, stride, dim, nch, out); break;
        case 3: hipLaunchKernelGGL((vg_minmax_kernel<VT, 3>), g, b, 0, stream, rows, n_rows, stride, dim, nch, out); break;
        case 4: hipLaunchKernelGGL((vg_minmax_kernel<VT, 4>), g, b, 0, stream, rows, n_rows, stride, dim, nch, out); break;
        case 6: hipLaunchKernelGGL((vg_minmax_kernel<VT, 6>), g, b, 0, stream, rows, n_rows, stride, dim, nch, out); break;
        default: hipLaunchKernelGGL((vg_minmax_kernel<VT, 8>), g, b, 0, stream, rows, n_rows, stride, dim, nch, out); break;
    }
}
template <int VT> static void vgq_quantize_go(int G, dim3 g, dim3 b, hipStream_t stream, const uint8_t *rows, long long row0, long long n_rows,
                                              long long stride, int dim, int nch, float scale, float offset, int qtype_u8, uint8_t *out) {
#define VGQ_LAUNCH(GG) hipLaunchKernelGGL((vg_quantize_kernel<VT, GG>), g, b, 0, stream, rows, row0, n_rows, stride, dim, nch, scale, offset, qtype_u8, out)
    switch (G) {
        case 1: VGQ_LAUNCH(1); break;
        case 2: VGQ_LAUNCH(2); break;
        case 3: VGQ_LAUNCH(3); break;
        case 4: VGQ_LAUNCH(4); break;
        case 6: VGQ_LAUNCH(6); break;
        default: VGQ_LAUNCH(8); break;
    }
#undef VGQ_LAUNCH
}

extern "C" int vg_quant_minmax_launch(const uint8_t *rows, long long n_rows, long long stride, int dim, int vtype,
                                      uint32_t *dev_out3, hipStream_t stream) {
    const uint32_t init[3] = {vg_f32_sortable(3.402823466e+38f), vg_f32_sortable(-3.402823466e+38f), 0u};
    hipError_t e = hipMemcpyAsync(dev_out3, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    const int nch = (int)(stride / 16), G = vgq_pick_g(nch);
    const dim3 g(std::min(vgq_blocks(n_rows), 256u * 8u)), b(256);          // (8 workgroups of 4 wavefronts per CU: full occupancy, 2048 atomic sets)
    switch (vtype) {
        case T_F32: vgq_minmax_go<T_F32>(G, g, b, stream, rows, n_rows, stride, dim, nch, dev_out3); break;
        case T_F16: vgq_minmax_go<T_F16>(G, g, b, stream, rows, n_rows, stride, dim, nch, dev_out3); break;
        case T_BF16: vgq_minmax_go<T_BF16>(G, g, b, stream, rows, n_rows, stride, dim, nch, dev_out3); break;
        case T_U8: vgq_minmax_go<T_U8>(G, g, b, stream, rows, n_rows, stride, dim, nch, dev_out3); break;
        default: vgq_minmax_go<T_I8>(G, g, b, stream, rows, n_rows, stride, dim, nch, dev_out3); break;
    }
    return (int)hipGetLastError();
}

extern "C" int vg_quant_quantize_launch(const uint8_t *rows, long long row0, long long n_rows, long long stride, int dim,
                                        int vtype, float scale, float offset, int qtype_u8, uint8_t *dev_out,
                                        hipStream_t stream) {
    const int nch = (int)(stride / 16), G = vgq_pick_g(nch);
    const dim3 g(vgq_blocks(n_rows)), b(256);
    switch (vtype) {
        case T_F32: vgq_quantize_go<T_F32>(G, g, b, stream, rows, row0, n_rows, stride, dim, nch, scale, offset, qtype_u8, dev_out); break;
        case T_F16: vgq_quantize_go<T_F16>(G, g, b, stream, rows, row0, n_rows, stride, dim, nch, scale, offset, qtype_u8, dev_out); break;
        case T_BF16: vgq_quantize_go<T_BF16>(G, g, b, stream, rows, row0, n_rows, stride, dim, nch, scale, offset, qtype_u8, dev_out); break;
        case T_U8: vgq_quantize_go<T_U8>(G, g, b, stream, rows, row0, n_rows, stride, dim, nch, scale, offset, qtype_u8, dev_out); break;
        default: vgq_quantize_go<T_I8>(G, g, b, stream, rows, row0, n_rows, stride, dim, nch, scale, offset, qtype_u8, dev_out); break;
    }
    return (int)hipGetLastError();
}
