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

#include "vg_device.h"
#include "vg_half.h"

__device__ inline float vgq_elem(int vtype, const uint8_t *row, int i) {
    switch (vtype) {
        case T_F32: return reinterpret_cast<const float *>(row)[i];
        case T_F16: return vg_h2f(reinterpret_cast<const uint16_t *>(row)[i]);
        case T_BF16: return vg_b2f(reinterpret_cast<const uint16_t *>(row)[i]);
        case T_U8: return (float)row[i];
        default: return (float)reinterpret_cast<const int8_t *>(row)[i];
    }
}

// out[0] = sortable(min), out[1] = sortable(max), out[2] = any negative.  Pre-set by the host to
// sortable(FLT_MAX), sortable(-FLT_MAX), 0 (the reference's initial values, :1197-1198).
__global__ __launch_bounds__(256) void vg_minmax_kernel(const uint8_t *rows, long long n_rows, long long stride, int dim,
                                                         int vtype, uint32_t *out) {
    const long long total = n_rows * (long long)dim;
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
    int neg = 0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / dim;
        const int i = (int)(e - r * dim);
        const float v = vgq_elem(vtype, rows + r * stride, i);
        if (v < lo) lo = v;
        if (v > hi) hi = v;
        if (v < 0.0f) neg = 1;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        const int n2 = __shfl_xor(neg, off);
        if (l2 < lo) lo = l2;
        if (h2 > hi) hi = h2;
        neg |= n2;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], vg_f32_sortable(lo));
        atomicMax(&out[1], vg_f32_sortable(hi));
        if (neg) atomicOr(&out[2], 1u);
    }
}

__device__ inline int vgq_trunc_x86(float r) {           // cvttss2si semantics
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) return (int)0x80000000u;
    return (int)r;
}

__global__ __launch_bounds__(256) void vg_quantize_kernel(const uint8_t *rows, long long row0, long long n_rows,
                                                           long long stride, int dim, int vtype, float scale, float offset,
                                                           int qtype_u8, uint8_t *out) {
    const long long total = n_rows * (long long)dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / dim;
        const int i = (int)(e - r * dim);
        const float v = vgq_elem(vtype, rows + (row0 + r) * stride, i);
        const float s = (v - offset) * scale;                              // two roundings, no contraction
        const float rr = s + 0.5f * (1.0f - 2.0f * (s < 0.0f ? 1.0f : 0.0f));
        uint8_t q;
        if (vtype == T_F32) {                                              // sqlite-vector.c:517-548, :626-656
            const int ir = vgq_trunc_x86(rr);
            if (qtype_u8) q = (uint8_t)(ir > 255 ? 255 : (ir < 0 ? 0 : ir));
            else q = (uint8_t)(int8_t)(ir > 127 ? 127 : (ir < -128 ? -128 : ir));
        } else if (qtype_u8) {                                             // q_round_u8, :495-504
            if (!isfinite(s)) q = (s > 0.0f) ? 255u : 0u;
            else if (rr >= 255.0f) q = 255u;
            else if (rr <= 0.0f) q = 0u;
            else q = (uint8_t)(int)rr;
        } else {                                                           // q_round_s8, :506-515
            int8_t t;
            if (!isfinite(s)) t = (s > 0.0f) ? 127 : (s < 0.0f ? -128 : 0);
            else if (rr >= 127.0f) t = 127;
            else if (rr <= -128.0f) t = -128;
            else t = (int8_t)(int)rr;
            q = (uint8_t)t;
        }
        out[e] = q;
    }
}

extern "C" int vg_quant_minmax_launch(const uint8_t *rows, long long n_rows, long long stride, int dim, int vtype,
                                      uint32_t *dev_out3, hipStream_t stream) {
    const uint32_t init[3] = {vg_f32_sortable(3.402823466e+38f), vg_f32_sortable(-3.402823466e+38f), 0u};
    hipError_t e = hipMemcpyAsync(dev_out3, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return (int)e;
    const long long total = n_rows * (long long)dim;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vg_minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows, n_rows, stride, dim, vtype, dev_out3);
    return (int)hipGetLastError();
}

extern "C" int vg_quant_quantize_launch(const uint8_t *rows, long long row0, long long n_rows, long long stride, int dim,
                                        int vtype, float scale, float offset, int qtype_u8, uint8_t *dev_out,
                                        hipStream_t stream) {
    const long long total = n_rows * (long long)dim;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vg_quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows, row0, n_rows, stride, dim,
                       vtype, scale, offset, qtype_u8, dev_out);
    return (int)hipGetLastError();
}
