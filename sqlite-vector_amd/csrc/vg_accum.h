// vg_accum.h - per-(element type, metric) lane accumulators for the scan kernel.
//
// A row is cut into 16-byte chunks; 2^lpr_log2 lanes cooperate on one row, each lane folding its chunks into
// an Accum, then a butterfly over the lane group produces the row total and finish() applies the reference's
// scalar epilogue.  Reference semantics followed (file:line under /root/reference/src/):
//   f32   distance-avx2.c:67-162 / distance-cpu.c:39-159   (f32 accumulation; summation ORDER differs, <=1e-5 rel)
//   u8    distance-avx2.c:586-753   exact 32-bit integer sums, ONE int->float conversion, float epilogue
//   i8    distance-avx2.c:757-950   same, signed
//   f16   distance-avx2.c:166-364   f32 difference / f32 product widened to f64, f64 accumulation
//   bf16  distance-avx2.c:368-582   f64 difference, f32 product widened to f64, f64 accumulation
#pragma once

#include "vg_device.h"
#include "vg_half.h"

template <int VT, int ACC> struct Accum;

// ============================================================================================ f32

template <int ACC> struct Accum<T_F32, ACC> {
    float a0, a1, a2, a3;   // L2: sum (q-x)^2 | COS/DOT: sum q*x | L1: sum |q-x|
    float n0, n1, n2, n3;   // COS: sum x*x
    struct QStat { float qq; };

    __device__ inline void init() { a0 = a1 = a2 = a3 = 0.0f; n0 = n1 = n2 = n3 = 0.0f; }

    __device__ inline void chunk(const uint4 &qv, const uint4 &xv) {
        const float q0 = __uint_as_float(qv.x), q1 = __uint_as_float(qv.y), q2 = __uint_as_float(qv.z), q3 = __uint_as_float(qv.w);
        const float x0 = __uint_as_float(xv.x), x1 = __uint_as_float(xv.y), x2 = __uint_as_float(xv.z), x3 = __uint_as_float(xv.w);
        if (ACC == A_L2) {
            const float d0 = q0 - x0, d1 = q1 - x1, d2 = q2 - x2, d3 = q3 - x3;
            a0 = fmaf(d0, d0, a0); a1 = fmaf(d1, d1, a1); a2 = fmaf(d2, d2, a2); a3 = fmaf(d3, d3, a3);
        } else if (ACC == A_L1) {
            a0 += fabsf(q0 - x0); a1 += fabsf(q1 - x1); a2 += fabsf(q2 - x2); a3 += fabsf(q3 - x3);
        } else {
            a0 = fmaf(q0, x0, a0); a1 = fmaf(q1, x1, a1); a2 = fmaf(q2, x2, a2); a3 = fmaf(q3, x3, a3);
            if (ACC == A_COS) {
                n0 = fmaf(x0, x0, n0); n1 = fmaf(x1, x1, n1); n2 = fmaf(x2, x2, n2); n3 = fmaf(x3, x3, n3);
            }
        }
    }

    // sum of squares of the query over this lane group (only cosine needs it)
    template <int U>
    __device__ static inline QStat query_stat(const uint4 (&q)[U], int lpr_log2) {
        QStat s; s.qq = 0.0f;
        if (ACC == A_COS) {
            Accum<T_F32, A_DOT> t; t.init();
#pragma unroll
            for (int u = 0; u < U; ++u) t.chunk(q[u], q[u]);
            s.qq = vg_group_sum((t.a0 + t.a1) + (t.a2 + t.a3), lpr_log2);
        }
        return s;
    }

    __device__ inline bool special(const QStat &, int) const { return false; }
    __device__ static inline void merge_qstat(QStat &into, const QStat &part) { into.qq += part.qq; }

    __device__ inline float finish(const QStat &qs, int lpr_log2, int root) {
        float a = vg_group_sum((a0 + a1) + (a2 + a3), lpr_log2);
        if (ACC == A_L2) return root ? sqrtf(a) : a;                       // distance-avx2.c:99
        if (ACC == A_L1) return a;                                         // :125
        if (ACC == A_DOT) return -a;                                       // :150
        float nb = vg_group_sum((n0 + n1) + (n2 + n3), lpr_log2);
        return vg_cosine_from_norms(a, sqrtf(qs.qq), sqrtf(nb));           // :153-162
    }
};

// ============================================================================================ u8 / i8

template <int VT>
__device__ inline uint32_t vg_dot4(uint32_t a, uint32_t b, uint32_t c) {
    if (VT == T_U8) return __builtin_amdgcn_udot4(a, b, c, false);
    return (uint32_t)__builtin_amdgcn_sdot4((int)a, (int)b, (int)c, false);
}

template <int VT, int ACC> struct AccumInt {
    uint32_t sqx;           // sum q*x  (DOT/COS/L2) | sum |q-x| (L1)       -- exact, modulo 2^32 like the reference
    uint32_t sxx;           // sum x*x  (COS/L2)
    struct QStat { uint32_t qq; };

    __device__ inline void init() { sqx = 0; sxx = 0; }

    __device__ inline void dword(uint32_t q, uint32_t x) {
        if (ACC == A_L1) {
            // |q-x| per byte: v_sad_u8; signed bytes are biased by 128 first (|a-b| is shift invariant)
            const uint32_t bias = (VT == T_I8) ? 0x80808080u : 0u;
            sqx = __builtin_amdgcn_sad_u8(q ^ bias, x ^ bias, sqx);
        } else {
            sqx = vg_dot4<VT>(q, x, sqx);
            if (ACC != A_DOT) sxx = vg_dot4<VT>(x, x, sxx);
        }
    }
    __device__ inline void chunk(const uint4 &qv, const uint4 &xv) {
        dword(qv.x, xv.x); dword(qv.y, xv.y); dword(qv.z, xv.z); dword(qv.w, xv.w);
    }

    template <int U>
    __device__ static inline QStat query_stat(const uint4 (&q)[U], int lpr_log2) {
        QStat s; s.qq = 0;
        if (ACC == A_L2 || ACC == A_COS) {
            uint32_t t = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                t = vg_dot4<VT>(q[u].x, q[u].x, t); t = vg_dot4<VT>(q[u].y, q[u].y, t);
                t = vg_dot4<VT>(q[u].z, q[u].z, t); t = vg_dot4<VT>(q[u].w, q[u].w, t);
            }
            s.qq = vg_group_sum(t, lpr_log2);
        }
        return s;
    }

    __device__ inline bool special(const QStat &, int) const { return false; }
    __device__ static inline void merge_qstat(QStat &into, const QStat &part) { into.qq += part.qq; }

    // int -> float the way the reference's totals convert: u8 totals are uint32_t everywhere, i8 L2 totals are
    // uint32_t (distance-avx2.c:816) while i8 dot / L1 totals are int32_t (:871, :925)
    __device__ static inline float as_float_signed_like(uint32_t v) {
        return (VT == T_U8) ? (float)v : (float)(int32_t)v;
    }

    __device__ inline float finish(const QStat &qs, int lpr_log2, int root) {
        const uint32_t qx = vg_group_sum(sqx, lpr_log2);
        if (ACC == A_L1) return as_float_signed_like(qx);
        if (ACC == A_DOT) return -as_float_signed_like(qx);
        const uint32_t xx = vg_group_sum(sxx, lpr_log2);
        if (ACC == A_L2) {
            // sum (q-x)^2 = qq + xx - 2 qx, exact in modular 32-bit arithmetic
            const uint32_t total = qs.qq + xx - 2u * qx;
            const float t = (float)total;
            return root ? sqrtf(t) : t;
        }
        const float dot = as_float_signed_like(qx);
        const float na = sqrtf(as_float_signed_like(qs.qq));
        const float nb = sqrtf(as_float_signed_like(xx));
        return vg_cosine_from_norms(dot, na, nb);
    }
};

template <int ACC> struct Accum<T_U8, ACC> : AccumInt<T_U8, ACC> {};
template <int ACC> struct Accum<T_I8, ACC> : AccumInt<T_I8, ACC> {};

// ============================================================================================ f16 / bf16 (vg_half.h)
template <int ACC> struct Accum<T_F16, ACC> : AccumHalf<T_F16, ACC> {};
template <int ACC> struct Accum<T_BF16, ACC> : AccumHalf<T_BF16, ACC> {};
