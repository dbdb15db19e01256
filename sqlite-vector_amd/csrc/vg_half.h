// vg_half.h - f16 / bf16 support for the scan kernel.
//
// Fast path (Accum<T_F16|T_BF16, ACC>): rows made of finite values.  Arithmetic follows distance-avx2.c:
//   f16  : f32 difference / f32 product (exact for halves), widened to f64, f64 accumulation   (:166-340)
//   bf16 : f64 difference; f32 product widened to f64; f64 accumulation                          (:368-569)
// Only the f64 summation ORDER differs from the 2 x 4-lane AVX2 accumulators, which is invisible after the final
// rounding to float except in the last ulp.
//
// Any Inf/NaN element (exponent field all ones) in the row - or in the query - sends that row to the SLOW path:
// the owning lane replays the reference's algorithm element by element, including its block(8)/tail split and
// its quirks, so special-value results are bit-identical to distance-avx2.c:
//   * Inf mismatch -> +Inf (L2/L1); NaN lanes contribute 0; same-signed Inf pair: 0 in a block, NaN in the tail;
//   * dot: the first Inf-involving lane (in element order) that is not Inf*0 decides +-Inf; bf16 blocks test Inf
//     BEFORE NaN and let Inf*0 poison the sum with NaN (distance-avx2.c:502-531), tails ignore such lanes;
//   * cosine: f16 any Inf -> 1.0; non-finite dot / non-positive norms -> 1.0; clamp to [-1, 1].
#pragma once

#include "vg_device.h"

typedef _Float16 vg_half2 __attribute__((ext_vector_type(2)));

__device__ inline float vg_h2f(uint32_t h16) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h16); }
__device__ inline float vg_b2f(uint32_t b16) { return __uint_as_float(b16 << 16); }

template <int VT> __device__ inline float vg_elem_f32(uint32_t bits16) { return VT == T_F16 ? vg_h2f(bits16) : vg_b2f(bits16); }
template <int VT> __device__ inline bool vg_is_inf16(uint32_t h) { return VT == T_F16 ? (h & 0x7FFFu) == 0x7C00u : (h & 0x7FFFu) == 0x7F80u; }
template <int VT> __device__ inline bool vg_is_nan16(uint32_t h) {
    return VT == T_F16 ? ((h & 0x7C00u) == 0x7C00u && (h & 0x03FFu)) : ((h & 0x7F80u) == 0x7F80u && (h & 0x007Fu));
}
__device__ inline bool vg_is_zero16(uint32_t h) { return (h & 0x7FFFu) == 0; }
__device__ inline uint32_t vg_sign16(uint32_t h) { return (h >> 15) & 1u; }

// nonzero iff either 16-bit half of w has an all-ones exponent (Inf or NaN)
template <int VT> __device__ inline uint32_t vg_special_pair(uint32_t w) {
    if (VT == T_F16) return ((w & 0x7C007C00u) + 0x04000400u) & 0x80008000u;
    return ((w & 0x7F807F80u) + 0x00800080u) & 0x80008000u;
}

// two packed 16-bit elements -> two floats
template <int VT> __device__ inline void vg_unpack2(uint32_t w, float &lo, float &hi) {
    if (VT == T_F16) {
        vg_half2 h = __builtin_bit_cast(vg_half2, w);
        lo = (float)h.x; hi = (float)h.y;
    } else {
        lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xFFFF0000u);
    }
}

// ============================================================================================ fast path

// f16: v_fma_mix_f32 reads its f16 operands straight out of the packed dword (op_sel picks the half, op_sel_hi marks a source as
// f16) and produces the f32 result with ONE rounding - (float)q - (float)x as fma(q, 1.0, -x), (float)q * (float)x as fma(q, x, 0)
// - bit for bit what the two conversions + the f32 operation compute, in one VALU instruction instead of three (the unpacking was
// 40 % of the f16 kernels' instruction stream; they are bound by it, not by HBM).  f16 denormal inputs are honoured by the mix
// instructions (unlike v_mad_mix); the f32 result follows the kernel's f32 denormal mode like v_sub_f32 / v_mul_f32 do.
#ifndef VG_HALF_FMA_MIX
#define VG_HALF_FMA_MIX 1
#endif
__device__ inline void vg_f16_diff2(uint32_t qw, uint32_t xw, float &d0, float &d1) {
    asm("v_fma_mix_f32 %0, %2, 1.0, -%3 op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %1, %2, 1.0, -%3 op_sel:[1,0,1] op_sel_hi:[1,0,1]"
        : "=&v"(d0), "=v"(d1) : "v"(qw), "v"(xw));
}
__device__ inline void vg_f16_prod2(uint32_t qw, uint32_t xw, float &p0, float &p1) {
    asm("v_fma_mix_f32 %0, %2, %3, 0 op_sel_hi:[1,1,0]\n\t"
        "v_fma_mix_f32 %1, %2, %3, 0 op_sel:[1,1,0] op_sel_hi:[1,1,0]"
        : "=&v"(p0), "=v"(p1) : "v"(qw), "v"(xw));
}

// Which rows hold an Inf / NaN element is read off the row's f64 total instead of being tested element pair by element pair (three
// VALU operations per pair, a quarter to a third of the fast path's instruction stream): every such element makes the total
// non-finite - x = NaN poisons its term; x = +-Inf gives q - x = -+Inf (or NaN) for L2 / L1 and q * x = +-Inf, or NaN for q = 0,
// for the products - and opposite infinities meet as NaN, so nothing cancels back to a finite value.  The converse only costs
// time: a row of finite bf16 values whose f32 product overflows also takes the slow path, which is the reference's algorithm
// itself.  (f64 never overflows on finite 16-bit inputs: |term| < 2^256.)
#ifndef VG_HALF_FLAG_FROM_SUM
#define VG_HALF_FLAG_FROM_SUM 1
#endif
__device__ inline uint32_t vg_f64_nonfinite(double s) { return ((uint32_t)__double2hiint(s) & 0x7FF00000u) == 0x7FF00000u ? 1u : 0u; }

template <int VT, int ACC> struct AccumHalf {
    // four independent f64 accumulators per lane: a single chain of dependent v_fma_f64 / v_add_f64 (8 per chunk)
    // left the kernel latency-bound at ~5 TB/s
    double a0, a1, a2, a3;  // L2: sum d^2 | L1: sum |d| | DOT/COS: sum q*x
    double n0, n1, n2, n3;  // COS: sum x*x
    uint32_t flag;          // set by finish(): the row's f64 total is non-finite, i.e. the row takes the exact slow path (the per-pair
                            // test of VG_HALF_FLAG_FROM_SUM=0 builds: nonzero once this lane saw an Inf / NaN element)
    struct QStat { double qq; uint32_t qspecial; };

    __device__ inline void init() { a0 = a1 = a2 = a3 = 0.0; n0 = n1 = n2 = n3 = 0.0; flag = 0; }

    __device__ inline void pair(uint32_t qw, uint32_t xw, double &s0, double &s1, double &m0, double &m1) {
        if constexpr (VG_HALF_FLAG_FROM_SUM == 0) flag |= vg_special_pair<VT>(xw);
        if constexpr (VT == T_F16 && VG_HALF_FMA_MIX != 0 && ACC != A_COS) {
            float e0, e1;
            if (ACC == A_L2) {                  // f32 subtract, square in f64 (distance-avx2.c:186-205)
                vg_f16_diff2(qw, xw, e0, e1);
                const double d0 = (double)e0, d1 = (double)e1;
                s0 = fma(d0, d0, s0); s1 = fma(d1, d1, s1);
            } else if (ACC == A_L1) {
                vg_f16_diff2(qw, xw, e0, e1);
                s0 += (double)fabsf(e0); s1 += (double)fabsf(e1);
            } else {                            // A_DOT / A_COSN: exact f32 products
                vg_f16_prod2(qw, xw, e0, e1);
                s0 += (double)e0; s1 += (double)e1;
            }
            return;
        }
        float q0, q1, x0, x1;
        vg_unpack2<VT>(qw, q0, q1);
        vg_unpack2<VT>(xw, x0, x1);
        if (ACC == A_L2) {
            if (VT == T_F16) {              // f32 subtract, square in f64 (distance-avx2.c:186-205)
                const double d0 = (double)(q0 - x0), d1 = (double)(q1 - x1);
                s0 = fma(d0, d0, s0); s1 = fma(d1, d1, s1);
            } else {
                // The reference subtracts in f64 (distance-avx2.c:383-409).  The f32 difference of two bf16 values (8 significant bits) is
                // EXACT unless their exponents lie more than 16 apart - then it is off by <= 2^-24 of itself, 2^-23 of its square, far
                // inside the bar of finite rows (2 ulp of the float result) - and it saves two of the four f32 -> f64 conversions per
                // element pair: the bf16 L2 kernel was the slowest plain scan (0.75-0.79 of the HBM peak against 0.82 for f16, whose
                // reference arithmetic is exactly this).  VG_BF16_F64_DIFF=1 keeps the f64 subtraction (A/B).
#if defined(VG_BF16_F64_DIFF) && VG_BF16_F64_DIFF
                const double d0 = (double)q0 - (double)x0, d1 = (double)q1 - (double)x1;
#else
                const double d0 = (double)(q0 - x0), d1 = (double)(q1 - x1);
#endif
                s0 = fma(d0, d0, s0); s1 = fma(d1, d1, s1);
            }
        } else if (ACC == A_L1) {
            if (VT == T_F16) { s0 += (double)fabsf(q0 - x0); s1 += (double)fabsf(q1 - x1); }
            else { s0 += fabs((double)q0 - (double)x0); s1 += fabs((double)q1 - (double)x1); }
        } else {                             // f32 product (exact for finite halves / bf16 unless it over/underflows)
            s0 += (double)(q0 * x0); s1 += (double)(q1 * x1);
            if (ACC == A_COS) { m0 += (double)(x0 * x0); m1 += (double)(x1 * x1); }      // A_COSN: cached per row
        }
    }
    __device__ inline void chunk(const uint4 &qv, const uint4 &xv) {
        pair(qv.x, xv.x, a0, a1, n0, n1); pair(qv.y, xv.y, a2, a3, n2, n3);
        pair(qv.z, xv.z, a0, a1, n0, n1); pair(qv.w, xv.w, a2, a3, n2, n3);
    }

    template <int U>
    __device__ static inline QStat query_stat(const uint4 (&q)[U], int lpr_log2) {
        QStat s; s.qq = 0.0;
        uint32_t sp = 0;
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sp |= vg_special_pair<VT>(w[j]);
                float lo, hi;
                vg_unpack2<VT>(w[j], lo, hi);
                t += (double)(lo * lo); t += (double)(hi * hi);
            }
        }
        s.qspecial = vg_group_or(sp, lpr_log2);
        if (ACC == A_COS || ACC == A_COSN) s.qq = vg_group_sum(t, lpr_log2);
        return s;
    }

    __device__ static inline void merge_qstat(QStat &into, const QStat &part) { into.qq += part.qq; into.qspecial |= part.qspecial; }

    // true for every lane of the group if the row (as seen by finish(), which must have run) or the query has a special element
    __device__ inline bool special(const QStat &qs, int lpr_log2) const {
        if constexpr (VG_HALF_FLAG_FROM_SUM != 0) return (flag | qs.qspecial) != 0;       // (finish() left the same total in every lane)
        return (vg_group_or(flag, lpr_log2) | qs.qspecial) != 0;
    }

    __device__ inline float finish(const QStat &qs, int lpr_log2, int root) {
        const double s = vg_group_sum((a0 + a1) + (a2 + a3), lpr_log2);
        if constexpr (VG_HALF_FLAG_FROM_SUM != 0) flag = vg_f64_nonfinite(s);
        if (ACC == A_L2) return root ? (float)sqrt(s) : (float)s;                 // distance-avx2.c:217, :421
        if (ACC == A_L1) return (float)s;
        if (ACC == A_DOT) return (float)(-s);
        const double nn = vg_group_sum((n0 + n1) + (n2 + n3), lpr_log2);
        return cosine_epilogue(qs, (float)s, (float)nn);
    }
    // A_COSN: the row's (float) sum x^2 comes from the corpus' cached vector (same f64 accumulation, done once per row)
    __device__ inline float finish_cached_norm(const QStat &qs, int lpr_log2, float nn_row) {
        const double s = vg_group_sum((a0 + a1) + (a2 + a3), lpr_log2);
        if constexpr (VG_HALF_FLAG_FROM_SUM != 0) flag = vg_f64_nonfinite(s);
        return cosine_epilogue(qs, (float)s, nn_row);
    }
    __device__ static inline float cosine_epilogue(const QStat &qs, float dot, float nnf) {
        // distance-avx2.c:343-364 / :571-582: float epilogue on the three rounded dot products
        const float na = sqrtf((float)qs.qq), nb = sqrtf(nnf);
        if (!(na > 0.0f) || !(nb > 0.0f) || !isfinite(na) || !isfinite(nb) || !isfinite(dot)) return 1.0f;
        float cs = __fdiv_rn(dot, na * nb);
        if (cs > 1.0f) cs = 1.0f;
        if (cs < -1.0f) cs = -1.0f;
        return 1.0f - cs;
    }
};

// ============================================================================================ slow path (one lane, exact)

__device__ inline double vg_hsum4(const double v[4]) { return (v[0] + v[2]) + (v[1] + v[3]); }   // hsum256d, :25-32

template <int VT> __device__ inline bool vg_inf_mismatch(uint32_t x, uint32_t y) {
    const bool xi = vg_is_inf16<VT>(x), yi = vg_is_inf16<VT>(y);
    return (xi || yi) && !(xi && yi && vg_sign16(x) == vg_sign16(y));
}

// L2 / L1 for f16 (distance-avx2.c:166-279) and bf16 (:368-489)
template <int VT, bool IS_L1>
__device__ inline float vg_slow_l2_l1(const uint16_t *a, const uint16_t *b, int n, int root) {
    double acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int k = 0; k < 8; ++k) if (vg_inf_mismatch<VT>(a[i + k], b[i + k])) return INFINITY;
        for (int k = 0; k < 8; ++k) {
            const uint32_t x = a[i + k], y = b[i + k];
            double d;
            if (VT == T_F16) {
                if (vg_is_nan16<VT>(x) || vg_is_nan16<VT>(y)) d = 0.0;
                else {
                    float df = vg_h2f(x) - vg_h2f(y);
                    if (IS_L1) df = fabsf(df);
                    d = (df != df) ? 0.0 : (double)df;
                }
            } else {
                d = (double)vg_b2f(x) - (double)vg_b2f(y);
                if (IS_L1) d = fabs(d);
                if (d != d) d = 0.0;
            }
            // lanes 0-3 -> first accumulator, lanes 4-7 -> second (L1 uses ONE accumulator: lane k then lane k+4)
            if (IS_L1) { acc0[k & 3] = acc0[k & 3] + d; }
            else if (k < 4) acc0[k] = acc0[k] + d * d;
            else acc1[k - 4] = acc1[k - 4] + d * d;
        }
    }
    double sum = IS_L1 ? vg_hsum4(acc0) : (vg_hsum4(acc0) + vg_hsum4(acc1));
    for (; i < n; ++i) {
        const uint32_t x = a[i], y = b[i];
        if (vg_inf_mismatch<VT>(x, y)) return INFINITY;
        if (vg_is_nan16<VT>(x) || vg_is_nan16<VT>(y)) continue;
        const double d = (double)vg_elem_f32<VT>(x) - (double)vg_elem_f32<VT>(y);
        if (IS_L1) sum += fabs(d); else sum = fma(d, d, sum);
    }
    if (IS_L1) return (float)sum;
    return root ? (float)sqrt(sum) : (float)sum;
}

// returns -dot like the reference (distance-avx2.c:281-340 f16, :491-569 bf16)
template <int VT>
__device__ inline float vg_slow_dot(const uint16_t *a, const uint16_t *b, int n) {
    double acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float pf[8];
        if (VT == T_F16) {
            for (int k = 0; k < 8; ++k) {
                const uint32_t x = a[i + k], y = b[i + k];
                if (vg_is_nan16<VT>(x) || vg_is_nan16<VT>(y)) { pf[k] = 0.0f; continue; }
                const bool xi = vg_is_inf16<VT>(x), yi = vg_is_inf16<VT>(y);
                if (xi || yi) {
                    if ((xi && vg_is_zero16(y)) || (yi && vg_is_zero16(x))) pf[k] = 0.0f;
                    else return (vg_sign16(x) ^ vg_sign16(y)) ? INFINITY : -INFINITY;
                } else {
                    const float p = vg_h2f(x) * vg_h2f(y);
                    if (isinf(p)) return (p > 0) ? -INFINITY : INFINITY;
                    pf[k] = (p != p) ? 0.0f : p;
                }
            }
        } else {
            for (int k = 0; k < 8; ++k) {                       // Inf test first; Inf*0 is only exempted from the return
                const uint32_t x = a[i + k], y = b[i + k];
                const bool xi = vg_is_inf16<VT>(x), yi = vg_is_inf16<VT>(y);
                if (xi || yi) {
                    if ((xi && vg_is_zero16(y)) || (yi && vg_is_zero16(x))) continue;
                    return (vg_sign16(x) ^ vg_sign16(y)) ? INFINITY : -INFINITY;
                }
            }
            for (int k = 0; k < 8; ++k) {
                float x = vg_b2f(a[i + k]), y = vg_b2f(b[i + k]);
                if (x != x) x = 0.0f;
                if (y != y) y = 0.0f;
                pf[k] = x * y;                                   // Inf * 0 -> NaN stays in the sum
            }
        }
        for (int k = 0; k < 4; ++k) { acc0[k] = acc0[k] + (double)pf[k]; acc1[k] = acc1[k] + (double)pf[k + 4]; }
    }
    double dot = vg_hsum4(acc0);
    dot += vg_hsum4(acc1);
    for (; i < n; ++i) {
        const uint32_t x = a[i], y = b[i];
        if (vg_is_nan16<VT>(x) || vg_is_nan16<VT>(y)) continue;
        const bool xi = vg_is_inf16<VT>(x), yi = vg_is_inf16<VT>(y);
        if (xi || yi) {
            if ((xi && vg_is_zero16(y)) || (yi && vg_is_zero16(x))) continue;
            return (vg_sign16(x) ^ vg_sign16(y)) ? INFINITY : -INFINITY;
        }
        const double p = (double)vg_elem_f32<VT>(x) * (double)vg_elem_f32<VT>(y);
        if (VT == T_F16) {
            if (isinf(p)) return (p > 0) ? -INFINITY : INFINITY;
            if (p == p) dot += p;
        } else {
            dot += p;
        }
    }
    return (float)(-dot);
}

template <int VT>
__device__ inline float vg_slow_cos(const uint16_t *a, const uint16_t *b, int n) {
    if (VT == T_F16)                                             // distance-avx2.c:347-350
        for (int i = 0; i < n; ++i) if (vg_is_inf16<VT>(a[i]) || vg_is_inf16<VT>(b[i])) return 1.0f;
    const float dot = -vg_slow_dot<VT>(a, b, n);
    const float na = sqrtf(-vg_slow_dot<VT>(a, a, n));
    const float nb = sqrtf(-vg_slow_dot<VT>(b, b, n));
    if (!(na > 0.0f) || !(nb > 0.0f) || !isfinite(na) || !isfinite(nb) || !isfinite(dot)) return 1.0f;
    float cs = __fdiv_rn(dot, na * nb);
    if (cs > 1.0f) cs = 1.0f;
    if (cs < -1.0f) cs = -1.0f;
    return 1.0f - cs;
}

template <int VT, int ACC>
__device__ inline float vg_slow_distance(const uint16_t *query, const uint16_t *row, int dim, int root) {
    if (ACC == A_L2) return vg_slow_l2_l1<VT, false>(query, row, dim, root);
    if (ACC == A_L1) return vg_slow_l2_l1<VT, true>(query, row, dim, 0);
    if (ACC == A_DOT) return vg_slow_dot<VT>(query, row, dim);
    return vg_slow_cos<VT>(query, row, dim);
}
