/*
 * oracle.c - CPU restatement of sqlite-vector's distance kernels, top-k slots and quantizer.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain C, no SIMD intrinsics: the "AVX2" backend
 * below replays distance-avx2.c's 8-lane accumulation ORDER with scalar arithmetic, so it is
 * bit-identical to the reference built with -mavx2 (no -mfma) on any host.
 *
 * Build with -ffp-contract=off: every multiply/add here must round separately exactly where
 * the reference's does; fused operations appear only where the reference calls fmaf()/fma().
 *
 * Citations: file:line relative to /root/reference/src/.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1              /* pthread_setaffinity_np, CPU_SET (the all-cores harness of the CPU baseline) */
#endif
#include "oracle.h"

#include <float.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ conversions */

static inline uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float float_of(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* IEEE binary16 -> binary32, exact (libs/fp16/fp16.h:115 computes the same value). */
float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    if (exp == 0x1F) return float_of(sign | 0x7F800000u | (man << 13));          /* Inf / NaN */
    if (exp != 0) return float_of(sign | ((exp + 112u) << 23) | (man << 13));     /* normal */
    if (man == 0) return float_of(sign);                                          /* +-0 */
    /* subnormal: man * 2^-24 */
    float v = (float)man * 0x1.0p-24f;
    return sign ? -v : v;
}

/* binary32 -> binary16, round-to-nearest-even (libs/fp16/fp16.h:256). */
uint16_t orc_f32_to_f16(float f) {
    uint32_t w = bits_of(f);
    uint16_t sign = (uint16_t)((w >> 16) & 0x8000u);
    uint32_t aw = w & 0x7FFFFFFFu;
    if (aw > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);                     /* NaN */
    if (aw >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);                    /* >= 65536 -> Inf (incl. Inf) */
    if (aw < 0x33000000u) return sign;                                           /* < 2^-25 -> 0 */
    int e = (int)(aw >> 23) - 127;
    uint32_t m = (aw & 0x7FFFFFu) | 0x800000u;                                    /* 24-bit significand */
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                                /* bits dropped */
    uint32_t kept = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (kept & 1u))) kept++;
    uint32_t out;
    if (e < -14) out = kept;                                                      /* subnormal (may carry into normal) */
    else out = ((uint32_t)(e + 15) << 10) + (kept - 0x400u);                      /* carry propagates into exponent */
    if (out >= 0x7C00u) out = 0x7C00u;
    return (uint16_t)(sign | out);
}

/* distance-cpu.h:100-102 */
float orc_bf16_to_f32(uint16_t h) { return float_of((uint32_t)h << 16); }

/* distance-cpu.h:103-108 */
uint16_t orc_f32_to_bf16(float f) {
    uint32_t x = bits_of(f);
    uint32_t round = 0x7FFFu + ((x >> 16) & 1u);
    return (uint16_t)((x + round) >> 16);
}

/* classifiers, distance-cpu.h:85-123 */
static inline int f16_nan(uint16_t h) { return (h & 0x7C00u) == 0x7C00u && (h & 0x03FFu); }
static inline int f16_inf(uint16_t h) { return (h & 0x7FFFu) == 0x7C00u; }
static inline int f16_zero(uint16_t h) { return (h & 0x7FFFu) == 0; }
static inline int bf_nan(uint16_t h) { return (h & 0x7F80u) == 0x7F80u && (h & 0x007Fu); }
static inline int bf_inf(uint16_t h) { return (h & 0x7FFFu) == 0x7F80u; }
static inline int bf_zero(uint16_t h) { return (h & 0x7FFFu) == 0; }
static inline int sgn16(uint16_t h) { return h >> 15; }

int orc_type_size(int type) {
    switch (type) {
        case ORC_TYPE_F32: return 4;
        case ORC_TYPE_F16: case ORC_TYPE_BF16: return 2;
        case ORC_TYPE_U8: case ORC_TYPE_I8: return 1;
    }
    return 0;
}

/* sqlite-vector.c:994-996 + its use at :2099 / :2141 */
float orc_clamp(float d) { return (fabsf(d) <= 8.0f * FLT_EPSILON) ? 0.0f : d; }

/* LASSQ step, distance-cpu.c:23-35 */
#define LASSQ(ad_) do { double ad__ = (ad_); if (ad__ != 0.0) {                       \
        if (scale < ad__) { double r__ = scale / ad__; ssq = 1.0 + ssq * (r__ * r__); scale = ad__; } \
        else { double r__ = ad__ / scale; ssq += r__ * r__; } } } while (0)

/* ------------------------------------------------------------------ CPU backend: f32 */

/* distance-cpu.c:39-72 */
static float cpu_f32_l2(const float *a, const float *b, int n, int root) {
    float s = 0.0f; int i = 0;
    for (; i + 4 <= n; i += 4) {
        float d0 = a[i] - b[i], d1 = a[i+1] - b[i+1], d2 = a[i+2] - b[i+2], d3 = a[i+3] - b[i+3];
        s += d0*d0 + d1*d1 + d2*d2 + d3*d3;
    }
    for (; i < n; ++i) { float d = a[i] - b[i]; s += d * d; }
    return root ? sqrtf(s) : s;
}

/* distance-cpu.c:74-110 */
static float cpu_f32_cos(const float *a, const float *b, int n) {
    float dot = 0.0f, nx = 0.0f, ny = 0.0f; int i = 0;
    for (; i + 4 <= n; i += 4) {
        float x0 = a[i], x1 = a[i+1], x2 = a[i+2], x3 = a[i+3];
        float y0 = b[i], y1 = b[i+1], y2 = b[i+2], y3 = b[i+3];
        dot += x0*y0 + x1*y1 + x2*y2 + x3*y3;
        nx  += x0*x0 + x1*x1 + x2*x2 + x3*x3;
        ny  += y0*y0 + y1*y1 + y2*y2 + y3*y3;
    }
    for (; i < n; ++i) { float x = a[i], y = b[i]; dot += x * y; nx += x * x; ny += y * y; }
    if (nx == 0.0f || ny == 0.0f) return 1.0f;
    return 1.0f - (dot / (sqrtf(nx) * sqrtf(ny)));
}

/* distance-cpu.c:112-136 */
static float cpu_f32_dot(const float *a, const float *b, int n) {
    float dot = 0.0f; int i = 0;
    for (; i + 4 <= n; i += 4)
        dot += a[i]*b[i] + a[i+1]*b[i+1] + a[i+2]*b[i+2] + a[i+3]*b[i+3];
    for (; i < n; ++i) dot += a[i] * b[i];
    return -dot;
}

/* distance-cpu.c:138-159 */
static float cpu_f32_l1(const float *a, const float *b, int n) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += fabsf(a[i] - b[i]);   /* reference adds one term at a time, in order */
    return s;
}

/* ------------------------------------------------------------------ CPU backend: bf16 */

/* distance-cpu.c:164-205 (f32 difference, LASSQ in double) */
static float cpu_bf16_l2(const uint16_t *a, const uint16_t *b, int n, int root) {
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; ++i) {
        float d = orc_bf16_to_f32(a[i]) - orc_bf16_to_f32(b[i]);
        if (isinf(d)) return INFINITY;
        if (!isnan(d)) LASSQ(fabs((double)d));
    }
    double ss = (scale == 0.0) ? 0.0 : scale * scale * ssq;
    return (float)(root ? sqrt(ss) : ss);
}

/* distance-cpu.c:207-254 (fmaf chains in element order) */
static float cpu_bf16_cos(const uint16_t *a, const uint16_t *b, int n) {
    float dot = 0.0f, nx = 0.0f, ny = 0.0f; int i = 0;
    for (; i + 4 <= n; i += 4) {
        float x[4], y[4];
        for (int k = 0; k < 4; ++k) { x[k] = orc_bf16_to_f32(a[i+k]); y[k] = orc_bf16_to_f32(b[i+k]); }
        for (int k = 0; k < 4; ++k) dot = fmaf(x[k], y[k], dot);
        for (int k = 0; k < 4; ++k) nx = fmaf(x[k], x[k], nx);
        for (int k = 0; k < 4; ++k) ny = fmaf(y[k], y[k], ny);
    }
    for (; i < n; ++i) {
        float x = orc_bf16_to_f32(a[i]), y = orc_bf16_to_f32(b[i]);
        dot = fmaf(x, y, dot); nx = fmaf(x, x, nx); ny = fmaf(y, y, ny);
    }
    if (nx == 0.0f || ny == 0.0f) return 1.0f;
    return 1.0f - (dot / (sqrtf(nx) * sqrtf(ny)));
}

/* distance-cpu.c:256-285 */
static float cpu_bf16_dot(const uint16_t *a, const uint16_t *b, int n) {
    float dot = 0.0f;
    for (int i = 0; i < n; ++i) dot = fmaf(orc_bf16_to_f32(a[i]), orc_bf16_to_f32(b[i]), dot);
    return -dot;
}

/* distance-cpu.c:287-314 */
static float cpu_bf16_l1(const uint16_t *a, const uint16_t *b, int n) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += fabsf(orc_bf16_to_f32(a[i]) - orc_bf16_to_f32(b[i]));
    return s;
}

/* ------------------------------------------------------------------ CPU backend: f16 */

static inline int f16_inf_mismatch(uint16_t x, uint16_t y) {
    int xi = f16_inf(x), yi = f16_inf(y);
    return (xi || yi) && !(xi && yi && sgn16(x) == sgn16(y));
}

/* distance-cpu.c:318-364: a 4-block returns +Inf if ANY of its lanes mismatches before any lane is added */
static float cpu_f16_l2(const uint16_t *a, const uint16_t *b, int n, int root) {
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; ++i) {
        if (f16_inf_mismatch(a[i], b[i])) return INFINITY;
        if (f16_nan(a[i]) || f16_nan(b[i])) continue;
        double d = (double)orc_f16_to_f32(a[i]) - (double)orc_f16_to_f32(b[i]);
        LASSQ(fabs(d));
    }
    double ss = (scale == 0.0) ? 0.0 : scale * scale * ssq;
    return (float)(root ? sqrt(ss) : ss);
}

/* distance-cpu.c:366-398 */
static float cpu_f16_l1(const uint16_t *a, const uint16_t *b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        if (f16_inf_mismatch(a[i], b[i])) return INFINITY;
        if (f16_nan(a[i]) || f16_nan(b[i])) continue;
        s += fabs((double)orc_f16_to_f32(a[i]) - (double)orc_f16_to_f32(b[i]));
    }
    return (float)s;
}

/* distance-cpu.c:400-430 */
static float cpu_f16_dot(const uint16_t *a, const uint16_t *b, int n) {
    double dot = 0.0;
    for (int i = 0; i < n; ++i) {
        float x = orc_f16_to_f32(a[i]), y = orc_f16_to_f32(b[i]);
        if (isnan(x) || isnan(y)) continue;
        double p = (double)x * (double)y;
        if (isinf(p)) return (p > 0) ? -INFINITY : INFINITY;
        dot += p;                                   /* Inf*0 = NaN is added: the scalar path does not mask it */
    }
    return (float)(-dot);
}

/* distance-cpu.c:432-466 */
static float cpu_f16_cos(const uint16_t *a, const uint16_t *b, int n) {
    double dot = 0.0, nx = 0.0, ny = 0.0;
    for (int i = 0; i < n; ++i) {
        float x = orc_f16_to_f32(a[i]), y = orc_f16_to_f32(b[i]);
        if (isnan(x) || isnan(y)) continue;
        if (isinf(x) || isinf(y)) return 1.0f;
        double xd = x, yd = y;
        dot += xd * yd; nx += xd * xd; ny += yd * yd;
    }
    double den = sqrt(nx) * sqrt(ny);
    if (!(den > 0.0) || !isfinite(den) || !isfinite(dot)) return 1.0f;
    double c = dot / den;
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    return (float)(1.0 - c);
}

/* ------------------------------------------------------------------ CPU backend: u8 / i8 */
/* distance-cpu.c:470-578 (u8) and :582-693 (i8): L2/dot/L1 accumulate in FLOAT, cosine in 32-bit ints */

#define CPU_INT_KERNELS(NAME, T, ACC_T)                                                        \
static float cpu_##NAME##_l2(const T *a, const T *b, int n, int root) {                        \
    float s = 0.0f; int i = 0;                                                                 \
    for (; i + 4 <= n; i += 4) {                                                               \
        int d0 = (int)a[i] - (int)b[i], d1 = (int)a[i+1] - (int)b[i+1];                        \
        int d2 = (int)a[i+2] - (int)b[i+2], d3 = (int)a[i+3] - (int)b[i+3];                    \
        s += (float)(d0*d0 + d1*d1 + d2*d2 + d3*d3);                                           \
    }                                                                                          \
    for (; i < n; ++i) { int d = (int)a[i] - (int)b[i]; s += (float)(d * d); }                 \
    return root ? sqrtf(s) : s;                                                                \
}                                                                                              \
static float cpu_##NAME##_cos(const T *a, const T *b, int n) {                                 \
    ACC_T dot = 0, na = 0, nb = 0;                                                             \
    for (int i = 0; i < n; ++i) {                                                              \
        ACC_T x = a[i], y = b[i]; dot += x * y; na += x * x; nb += y * y;                      \
    }                                                                                          \
    if (na == 0 || nb == 0) return 1.0f;                                                       \
    float cs = dot / (sqrtf((float)na) * sqrtf((float)nb));                                    \
    return 1.0f - cs;                                                                          \
}                                                                                              \
static float cpu_##NAME##_dot(const T *a, const T *b, int n) {                                 \
    float dot = 0.0f;                                                                          \
    for (int i = 0; i < n; ++i) dot += (float)a[i] * b[i];                                     \
    return -dot;                                                                               \
}                                                                                              \
static float cpu_##NAME##_l1(const T *a, const T *b, int n) {                                  \
    float s = 0.0f;                                                                            \
    for (int i = 0; i < n; ++i) s += fabsf((float)a[i] - (float)b[i]);                         \
    return s;                                                                                  \
}

CPU_INT_KERNELS(u8, uint8_t, uint32_t)
CPU_INT_KERNELS(i8, int8_t, int32_t)

/* ------------------------------------------------------------------ AVX2 backend: f32 */
/* 8 independent lane accumulators, lanes summed left to right, then a scalar tail. */

/* distance-avx2.c:67-100 */
static float avx_f32_l2(const float *a, const float *b, int n, int root) {
    float acc[8] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) { float d = a[i+l] - b[i+l]; acc[l] = acc[l] + d * d; }
    float t = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    for (; i < n; ++i) { float d = a[i] - b[i]; t += d * d; }
    return root ? sqrtf(t) : t;
}

/* distance-avx2.c:102-126 */
static float avx_f32_l1(const float *a, const float *b, int n) {
    float acc[8] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) acc[l] = acc[l] + fabsf(a[i+l] - b[i+l]);
    float t = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    for (; i < n; ++i) t += fabsf(a[i] - b[i]);
    return t;
}

/* distance-avx2.c:128-151: returns -dot */
static float avx_f32_dot(const float *a, const float *b, int n) {
    float acc[8] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) acc[l] = acc[l] + a[i+l] * b[i+l];
    float t = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    for (; i < n; ++i) t += a[i] * b[i];
    return -t;
}

/* distance-avx2.c:153-162: three dot passes */
static float avx_f32_cos(const float *a, const float *b, int n) {
    float dot = -avx_f32_dot(a, b, n);
    float na = sqrtf(-avx_f32_dot(a, a, n));
    float nb = sqrtf(-avx_f32_dot(b, b, n));
    if (na == 0.0f || nb == 0.0f) return 1.0f;
    return 1.0f - dot / (na * nb);
}

/* hsum256d, distance-avx2.c:25-32: (v0+v2) + (v1+v3) */
static inline double hsum4(const double v[4]) { return (v[0] + v[2]) + (v[1] + v[3]); }

/* ------------------------------------------------------------------ AVX2 backend: f16 */

/* distance-avx2.c:166-225 (non-FMA branch: mul then add) */
static float avx_f16_l2(const uint16_t *a, const uint16_t *b, int n, int root) {
    double acc0[4] = {0}, acc1[4] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int k = 0; k < 8; ++k) {                       /* block_has_l2_inf_mismatch_8, :35-44 */
            int xi = f16_inf(a[i+k]), yi = f16_inf(b[i+k]);
            if ((xi ^ yi) || (xi && yi && sgn16(a[i+k]) != sgn16(b[i+k]))) return INFINITY;
        }
        float df[8];
        for (int k = 0; k < 8; ++k) {
            if (f16_nan(a[i+k]) || f16_nan(b[i+k])) { df[k] = 0.0f; continue; }
            float d = orc_f16_to_f32(a[i+k]) - orc_f16_to_f32(b[i+k]);   /* f32 subtract */
            df[k] = isnan(d) ? 0.0f : d;
        }
        for (int k = 0; k < 4; ++k) {
            double lo = df[k], hi = df[k+4];
            acc0[k] = acc0[k] + lo * lo;
            acc1[k] = acc1[k] + hi * hi;
        }
    }
    double sum = hsum4(acc0) + hsum4(acc1);
    for (; i < n; ++i) {
        if (f16_inf_mismatch(a[i], b[i])) return INFINITY;
        if (f16_nan(a[i]) || f16_nan(b[i])) continue;
        double d = (double)orc_f16_to_f32(a[i]) - (double)orc_f16_to_f32(b[i]);
        sum = fma(d, d, sum);
    }
    return root ? (float)sqrt(sum) : (float)sum;
}

/* distance-avx2.c:235-279: one 4-lane accumulator takes lanes 0-3 then lanes 4-7 */
static float avx_f16_l1(const uint16_t *a, const uint16_t *b, int n) {
    double acc[4] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int k = 0; k < 8; ++k) {
            int xi = f16_inf(a[i+k]), yi = f16_inf(b[i+k]);
            if ((xi ^ yi) || (xi && yi && sgn16(a[i+k]) != sgn16(b[i+k]))) return INFINITY;
        }
        float df[8];
        for (int k = 0; k < 8; ++k) {
            if (f16_nan(a[i+k]) || f16_nan(b[i+k])) { df[k] = 0.0f; continue; }
            float d = fabsf(orc_f16_to_f32(a[i+k]) - orc_f16_to_f32(b[i+k]));
            df[k] = isnan(d) ? 0.0f : d;
        }
        for (int k = 0; k < 4; ++k) { acc[k] = acc[k] + (double)df[k]; acc[k] = acc[k] + (double)df[k+4]; }
    }
    double sum = hsum4(acc);
    for (; i < n; ++i) {
        if (f16_inf_mismatch(a[i], b[i])) return INFINITY;
        if (f16_nan(a[i]) || f16_nan(b[i])) continue;
        sum += fabs((double)orc_f16_to_f32(a[i]) - (double)orc_f16_to_f32(b[i]));
    }
    return (float)sum;
}

/* distance-avx2.c:281-340: returns -dot; f32 product widened to f64 */
static float avx_f16_dot(const uint16_t *a, const uint16_t *b, int n) {
    double acc0[4] = {0}, acc1[4] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8) {
        float pf[8];
        for (int k = 0; k < 8; ++k) {
            uint16_t x = a[i+k], y = b[i+k];
            if (f16_nan(x) || f16_nan(y)) { pf[k] = 0.0f; continue; }
            int xi = f16_inf(x), yi = f16_inf(y);
            if (xi || yi) {
                if ((xi && f16_zero(y)) || (yi && f16_zero(x))) pf[k] = 0.0f;     /* Inf*0 ignored */
                else return (sgn16(x) ^ sgn16(y)) ? INFINITY : -INFINITY;          /* first such lane decides */
            } else {
                float p = orc_f16_to_f32(x) * orc_f16_to_f32(y);
                if (isinf(p)) return (p > 0) ? -INFINITY : INFINITY;
                pf[k] = isnan(p) ? 0.0f : p;
            }
        }
        for (int k = 0; k < 4; ++k) { acc0[k] = acc0[k] + (double)pf[k]; acc1[k] = acc1[k] + (double)pf[k+4]; }
    }
    double dot = hsum4(acc0) + hsum4(acc1);
    for (; i < n; ++i) {
        uint16_t x = a[i], y = b[i];
        if (f16_nan(x) || f16_nan(y)) continue;
        int xi = f16_inf(x), yi = f16_inf(y);
        if (xi || yi) {
            if ((xi && f16_zero(y)) || (yi && f16_zero(x))) continue;
            return (sgn16(x) ^ sgn16(y)) ? INFINITY : -INFINITY;
        }
        double p = (double)orc_f16_to_f32(x) * (double)orc_f16_to_f32(y);
        if (isinf(p)) return (p > 0) ? -INFINITY : INFINITY;
        if (!isnan(p)) dot += p;
    }
    return (float)(-dot);
}

/* distance-avx2.c:343-364 */
static float avx_f16_cos(const uint16_t *a, const uint16_t *b, int n) {
    for (int i = 0; i < n; ++i) if (f16_inf(a[i]) || f16_inf(b[i])) return 1.0f;
    float dot = -avx_f16_dot(a, b, n);
    float na = sqrtf(-avx_f16_dot(a, a, n));
    float nb = sqrtf(-avx_f16_dot(b, b, n));
    if (!(na > 0.0f) || !(nb > 0.0f) || !isfinite(na) || !isfinite(nb) || !isfinite(dot)) return 1.0f;
    float c = dot / (na * nb);
    if (c > 1.0f) c = 1.0f;
    if (c < -1.0f) c = -1.0f;
    return 1.0f - c;
}

/* ------------------------------------------------------------------ AVX2 backend: bf16 */

static inline int bf_inf_mismatch(uint16_t x, uint16_t y) {
    int xi = bf_inf(x), yi = bf_inf(y);
    return (xi || yi) && !(xi && yi && sgn16(x) == sgn16(y));
}

/* distance-avx2.c:368-421: f64 subtract, NaN difference -> 0 */
static float avx_bf16_l2(const uint16_t *a, const uint16_t *b, int n, int root) {
    double acc0[4] = {0}, acc1[4] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int k = 0; k < 8; ++k) if (bf_inf_mismatch(a[i+k], b[i+k])) return INFINITY;
        for (int k = 0; k < 4; ++k) {
            double lo = (double)orc_bf16_to_f32(a[i+k]) - (double)orc_bf16_to_f32(b[i+k]);
            double hi = (double)orc_bf16_to_f32(a[i+k+4]) - (double)orc_bf16_to_f32(b[i+k+4]);
            if (lo != lo) lo = 0.0;
            if (hi != hi) hi = 0.0;
            acc0[k] = acc0[k] + lo * lo;
            acc1[k] = acc1[k] + hi * hi;
        }
    }
    double sum = hsum4(acc0) + hsum4(acc1);
    for (; i < n; ++i) {
        if (bf_inf_mismatch(a[i], b[i])) return INFINITY;
        if (bf_nan(a[i]) || bf_nan(b[i])) continue;
        double d = (double)orc_bf16_to_f32(a[i]) - (double)orc_bf16_to_f32(b[i]);
        sum = fma(d, d, sum);
    }
    return root ? (float)sqrt(sum) : (float)sum;
}

/* distance-avx2.c:431-489 */
static float avx_bf16_l1(const uint16_t *a, const uint16_t *b, int n) {
    double acc[4] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int k = 0; k < 8; ++k) if (bf_inf_mismatch(a[i+k], b[i+k])) return INFINITY;
        for (int k = 0; k < 4; ++k) {
            double lo = fabs((double)orc_bf16_to_f32(a[i+k]) - (double)orc_bf16_to_f32(b[i+k]));
            double hi = fabs((double)orc_bf16_to_f32(a[i+k+4]) - (double)orc_bf16_to_f32(b[i+k+4]));
            if (lo != lo) lo = 0.0;
            if (hi != hi) hi = 0.0;
            acc[k] = acc[k] + lo;
            acc[k] = acc[k] + hi;
        }
    }
    double sum = hsum4(acc);
    for (; i < n; ++i) {
        if (bf_inf_mismatch(a[i], b[i])) return INFINITY;
        if (bf_nan(a[i]) || bf_nan(b[i])) continue;
        sum += fabs((double)orc_bf16_to_f32(a[i]) - (double)orc_bf16_to_f32(b[i]));
    }
    return (float)sum;
}

/* distance-avx2.c:491-569.  Quirk kept: in the 8-blocks an Inf*0 lane is only exempted from the
 * early return; the vector multiply still produces NaN for it (only NaN INPUTS are zeroed, :522-525),
 * so the sum becomes NaN.  The scalar tail (:552-566) really ignores such lanes. */
static float avx_bf16_dot(const uint16_t *a, const uint16_t *b, int n) {
    double acc0[4] = {0}, acc1[4] = {0}; int i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int k = 0; k < 8; ++k) {
            uint16_t x = a[i+k], y = b[i+k];
            int xi = bf_inf(x), yi = bf_inf(y);
            if (xi || yi) {
                if ((xi && bf_zero(y)) || (yi && bf_zero(x))) continue;
                return (sgn16(x) ^ sgn16(y)) ? INFINITY : -INFINITY;
            }
        }
        float pf[8];
        for (int k = 0; k < 8; ++k) {
            float x = orc_bf16_to_f32(a[i+k]), y = orc_bf16_to_f32(b[i+k]);
            if (x != x) x = 0.0f;
            if (y != y) y = 0.0f;
            pf[k] = x * y;                                   /* f32 product */
        }
        for (int k = 0; k < 4; ++k) { acc0[k] = acc0[k] + (double)pf[k]; acc1[k] = acc1[k] + (double)pf[k+4]; }
    }
    double dot = hsum4(acc0);
    dot += hsum4(acc1);
    for (; i < n; ++i) {
        uint16_t x = a[i], y = b[i];
        if (bf_nan(x) || bf_nan(y)) continue;
        int xi = bf_inf(x), yi = bf_inf(y);
        if (xi || yi) {
            if ((xi && bf_zero(y)) || (yi && bf_zero(x))) continue;
            return (sgn16(x) ^ sgn16(y)) ? INFINITY : -INFINITY;
        }
        dot += (double)orc_bf16_to_f32(x) * (double)orc_bf16_to_f32(y);
    }
    return (float)(-dot);
}

/* distance-avx2.c:571-582 */
static float avx_bf16_cos(const uint16_t *a, const uint16_t *b, int n) {
    float dot = -avx_bf16_dot(a, b, n);
    float na = sqrtf(-avx_bf16_dot(a, a, n));
    float nb = sqrtf(-avx_bf16_dot(b, b, n));
    if (!(na > 0.0f) || !(nb > 0.0f) || !isfinite(na) || !isfinite(nb) || !isfinite(dot)) return 1.0f;
    float c = dot / (na * nb);
    if (c > 1.0f) c = 1.0f;
    if (c < -1.0f) c = -1.0f;
    return 1.0f - c;
}

/* ------------------------------------------------------------------ AVX2 backend: u8 / i8 */
/* distance-avx2.c:586-753 (u8), :757-950 (i8): exact 32-bit integer accumulation (order-free),
 * one int->float conversion at the end.  Totals are uint32 (wrapping) except i8 dot / i8 L1 (int32). */

static uint32_t u8_sum_sqdiff(const uint8_t *a, const uint8_t *b, int n) {
    uint32_t t = 0; for (int i = 0; i < n; ++i) { int d = (int)a[i] - (int)b[i]; t += (uint32_t)(d * d); } return t;
}
static uint32_t u8_sum_prod(const uint8_t *a, const uint8_t *b, int n) {
    uint32_t t = 0; for (int i = 0; i < n; ++i) t += (uint32_t)a[i] * (uint32_t)b[i]; return t;
}
static uint32_t u8_sum_absdiff(const uint8_t *a, const uint8_t *b, int n) {
    uint32_t t = 0; for (int i = 0; i < n; ++i) t += (uint32_t)abs((int)a[i] - (int)b[i]); return t;
}
static uint32_t i8_sum_sqdiff(const int8_t *a, const int8_t *b, int n) {
    uint32_t t = 0; for (int i = 0; i < n; ++i) { int d = (int)a[i] - (int)b[i]; t += (uint32_t)(d * d); } return t;
}
static int32_t i8_sum_prod(const int8_t *a, const int8_t *b, int n) {
    uint32_t t = 0; for (int i = 0; i < n; ++i) t += (uint32_t)((int)a[i] * (int)b[i]); return (int32_t)t;
}
static int32_t i8_sum_absdiff(const int8_t *a, const int8_t *b, int n) {
    uint32_t t = 0; for (int i = 0; i < n; ++i) t += (uint32_t)abs((int)a[i] - (int)b[i]); return (int32_t)t;
}

static float avx_u8_l2(const uint8_t *a, const uint8_t *b, int n, int root) {      /* :586-658 */
    float t = (float)u8_sum_sqdiff(a, b, n); return root ? sqrtf(t) : t;
}
static float avx_u8_dot(const uint8_t *a, const uint8_t *b, int n) { return -(float)u8_sum_prod(a, b, n); }   /* :660-700 */
static float avx_u8_l1(const uint8_t *a, const uint8_t *b, int n) { return (float)u8_sum_absdiff(a, b, n); }  /* :702-742 */
static float avx_u8_cos(const uint8_t *a, const uint8_t *b, int n) {                                          /* :744-753 */
    float dot = -avx_u8_dot(a, b, n);
    float na = sqrtf(-avx_u8_dot(a, a, n)), nb = sqrtf(-avx_u8_dot(b, b, n));
    if (na == 0.0f || nb == 0.0f) return 1.0f;
    return 1.0f - dot / (na * nb);
}
static float avx_i8_l2(const int8_t *a, const int8_t *b, int n, int root) {        /* :757-831 */
    float t = (float)i8_sum_sqdiff(a, b, n); return root ? sqrtf(t) : t;
}
static float avx_i8_dot(const int8_t *a, const int8_t *b, int n) { return -(float)i8_sum_prod(a, b, n); }     /* :833-885 */
static float avx_i8_l1(const int8_t *a, const int8_t *b, int n) { return (float)i8_sum_absdiff(a, b, n); }    /* :887-939 */
static float avx_i8_cos(const int8_t *a, const int8_t *b, int n) {                                            /* :941-950 */
    float dot = -avx_i8_dot(a, b, n);
    float na = sqrtf(-avx_i8_dot(a, a, n)), nb = sqrtf(-avx_i8_dot(b, b, n));
    if (na == 0.0f || nb == 0.0f) return 1.0f;
    return 1.0f - dot / (na * nb);
}

/* ------------------------------------------------------------------ dispatch */
/* restates dispatch_distance_table[metric][type] (distance-cpu.c:755-795, distance-avx2.c:956-990) */

float orc_distance(int backend, int metric, int type, const void *v1, const void *v2, int n) {
    int root = (metric == ORC_DIST_L2);
    if (backend == ORC_BACKEND_CPU) {
        switch (type) {
        case ORC_TYPE_F32:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return cpu_f32_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return cpu_f32_cos(v1, v2, n);
            case ORC_DIST_DOT: return cpu_f32_dot(v1, v2, n);
            case ORC_DIST_L1: return cpu_f32_l1(v1, v2, n);
            } break;
        case ORC_TYPE_F16:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return cpu_f16_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return cpu_f16_cos(v1, v2, n);
            case ORC_DIST_DOT: return cpu_f16_dot(v1, v2, n);
            case ORC_DIST_L1: return cpu_f16_l1(v1, v2, n);
            } break;
        case ORC_TYPE_BF16:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return cpu_bf16_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return cpu_bf16_cos(v1, v2, n);
            case ORC_DIST_DOT: return cpu_bf16_dot(v1, v2, n);
            case ORC_DIST_L1: return cpu_bf16_l1(v1, v2, n);
            } break;
        case ORC_TYPE_U8:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return cpu_u8_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return cpu_u8_cos(v1, v2, n);
            case ORC_DIST_DOT: return cpu_u8_dot(v1, v2, n);
            case ORC_DIST_L1: return cpu_u8_l1(v1, v2, n);
            } break;
        case ORC_TYPE_I8:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return cpu_i8_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return cpu_i8_cos(v1, v2, n);
            case ORC_DIST_DOT: return cpu_i8_dot(v1, v2, n);
            case ORC_DIST_L1: return cpu_i8_l1(v1, v2, n);
            } break;
        }
    } else {
        switch (type) {
        case ORC_TYPE_F32:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return avx_f32_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return avx_f32_cos(v1, v2, n);
            case ORC_DIST_DOT: return avx_f32_dot(v1, v2, n);
            case ORC_DIST_L1: return avx_f32_l1(v1, v2, n);
            } break;
        case ORC_TYPE_F16:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return avx_f16_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return avx_f16_cos(v1, v2, n);
            case ORC_DIST_DOT: return avx_f16_dot(v1, v2, n);
            case ORC_DIST_L1: return avx_f16_l1(v1, v2, n);
            } break;
        case ORC_TYPE_BF16:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return avx_bf16_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return avx_bf16_cos(v1, v2, n);
            case ORC_DIST_DOT: return avx_bf16_dot(v1, v2, n);
            case ORC_DIST_L1: return avx_bf16_l1(v1, v2, n);
            } break;
        case ORC_TYPE_U8:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return avx_u8_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return avx_u8_cos(v1, v2, n);
            case ORC_DIST_DOT: return avx_u8_dot(v1, v2, n);
            case ORC_DIST_L1: return avx_u8_l1(v1, v2, n);
            } break;
        case ORC_TYPE_I8:
            switch (metric) {
            case ORC_DIST_L2: case ORC_DIST_SQUARED_L2: return avx_i8_l2(v1, v2, n, root);
            case ORC_DIST_COSINE: return avx_i8_cos(v1, v2, n);
            case ORC_DIST_DOT: return avx_i8_dot(v1, v2, n);
            case ORC_DIST_L1: return avx_i8_l1(v1, v2, n);
            } break;
        }
    }
    return NAN;
}

/* ------------------------------------------------------------------ scan drivers */

/* per-row body of vFullScanRun (sqlite-vector.c:2098-2099) / vQuantRunMemory (:2140-2141) */
void orc_scan_distances(int backend, int metric, int type, const void *query,
                        const void *rows, int64_t n_rows, int64_t row_stride_bytes, int dim,
                        float *out_dist) {
    const uint8_t *p = (const uint8_t *)rows;
    for (int64_t r = 0; r < n_rows; ++r)
        out_dist[r] = orc_clamp(orc_distance(backend, metric, type, query, p + r * row_stride_bytes, dim));
}

/* vFullScanFindMaxIndex, sqlite-vector.c:2022-2049: index of the FIRST maximum (both branches) */
static int first_max(const double *v, int n) {
    int m = 0;
    for (int i = 1; i < n; ++i) if (v[i] > v[m]) m = i;
    return m;
}

/* vFullScanSortSlots, sqlite-vector.c:2051-2069: exchange sort; the INFINITY test on slot i happens
 * before slot i is finalised, exactly like the reference. */
static int sort_slots(double *d, int64_t *ids, int k) {
    int empty = 0;
    for (int i = 0; i + 1 < k; ++i) {
        if (d[i] == INFINITY) ++empty;
        for (int j = i + 1; j < k; ++j)
            if (d[j] < d[i]) {
                double td = d[i]; d[i] = d[j]; d[j] = td;
                int64_t ti = ids[i]; ids[i] = ids[j]; ids[j] = ti;
            }
    }
    if (d[k - 1] == INFINITY) ++empty;
    return empty;
}

int orc_topk_reference(const float *dist, const int64_t *rowids, int64_t n, int k,
                       int64_t *out_rowids, double *out_dist) {
    if (k <= 0) return 0;
    for (int i = 0; i < k; ++i) { out_rowids[i] = 0; out_dist[i] = INFINITY; }   /* :1808-1809 */
    int worst = 0;                                                                /* cursor zero-filled, :1887 */
    for (int64_t r = 0; r < n; ++r) {
        double d = (double)dist[r];
        if (d < out_dist[worst]) {                                                /* :2102 strict less-than */
            out_dist[worst] = d;
            out_rowids[worst] = rowids ? rowids[r] : (r + 1);
            worst = first_max(out_dist, k);
        }
    }
    return k - sort_slots(out_dist, out_rowids, k);                               /* :1816-1817 */
}

typedef struct { float d; int64_t pos; } cand_t;
static int cand_cmp(const void *x, const void *y) {
    const cand_t *a = (const cand_t *)x, *b = (const cand_t *)y;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->pos < b->pos) ? -1 : (a->pos > b->pos);
}

int orc_topk_ordered(const float *dist, const int64_t *rowids, int64_t n, int k,
                     int64_t *out_rowids, double *out_dist, int64_t *out_pos) {
    if (k <= 0) return 0;
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    int64_t m = 0;
    for (int64_t r = 0; r < n; ++r)
        if (dist[r] < INFINITY) { c[m].d = dist[r]; c[m].pos = r; ++m; }          /* NaN / +Inf never enter (:2102) */
    qsort(c, (size_t)m, sizeof(cand_t), cand_cmp);
    int cnt = (int)(m < k ? m : k);
    for (int i = 0; i < cnt; ++i) {
        out_dist[i] = (double)c[i].d;
        out_rowids[i] = rowids ? rowids[c[i].pos] : (c[i].pos + 1);
        if (out_pos) out_pos[i] = c[i].pos;
    }
    free(c);
    return cnt;
}

int orc_scan_topk_reference(int backend, int metric, int type, const void *query,
                            const void *rows, int64_t n_rows, int64_t row_stride_bytes, int dim,
                            const int64_t *rowids, int k, int64_t *out_rowids, double *out_dist) {
    if (k <= 0) return 0;
    const uint8_t *p = (const uint8_t *)rows;
    for (int i = 0; i < k; ++i) { out_rowids[i] = 0; out_dist[i] = INFINITY; }
    int worst = 0;
    double cur = out_dist[worst];
    for (int64_t r = 0; r < n_rows; ++r) {
        float d = orc_clamp(orc_distance(backend, metric, type, query, p + r * row_stride_bytes, dim));
        if (d < cur) {
            out_dist[worst] = d;
            out_rowids[worst] = rowids ? rowids[r] : (r + 1);
            worst = first_max(out_dist, k);
            cur = out_dist[worst];
        }
    }
    return k - sort_slots(out_dist, out_rowids, k);
}

int orc_scan_topk_with_fn(orc_distance_fn fn, const void *query, const void *rows, int64_t n_rows,
                          int64_t row_stride_bytes, int dim, const int64_t *rowids, int k,
                          int64_t *out_rowids, double *out_dist) {
    if (k <= 0) return 0;
    const uint8_t *p = (const uint8_t *)rows;
    for (int i = 0; i < k; ++i) { out_rowids[i] = 0; out_dist[i] = INFINITY; }
    int worst = 0;
    double cur = out_dist[worst];
    for (int64_t r = 0; r < n_rows; ++r) {
        float d = orc_clamp(fn(query, p + r * row_stride_bytes, dim));
        if (d < cur) {
            out_dist[worst] = d;
            out_rowids[worst] = rowids ? rowids[r] : (r + 1);
            worst = first_max(out_dist, k);
            cur = out_dist[worst];
        }
    }
    return k - sort_slots(out_dist, out_rowids, k);
}

/* ------------------------------------------------------------------ the reference's scan on every host core (bench.py's cpu_baseline.all_cores)
 *
 * The reference is single-threaded: one vFullScanRun walks the table (sqlite-vector.c:2089-2107).  The generous upper bound SURVEY 8(d)
 * asks for is an embarrassingly parallel ROW SPLIT of one corpus: thread i runs that loop - the reference's own kernel (fn, out of its
 * dispatch table: distance-avx2.c:67-100 for f32 L2) inside the slot loop above - over rows [i P, (i+1) P), query after query, for
 * `seconds`; a query's answer would be the k-way merge of the per-thread lists (the caller checks that once, untimed).  Plain pthreads:
 * no interpreter lock anywhere near the timed loop (round 4 drove this from Python threads: 4.2x one core on 256 threads). */
typedef struct {
    orc_distance_fn fn;
    const uint8_t *queries;          /* nq queries of query_stride bytes, taken round robin */
    int nq;
    int64_t query_stride;
    const uint8_t *rows;
    int64_t n_rows, row_stride;
    int dim, k;
    volatile int *go, *stop, *ready;
    int64_t scans;                   /* completed scans of this thread's range while the clock ran */
    int64_t *first_ids;              /* k: the list of query 0 over this range (positions 1-based within the range) */
    double *first_dist;
    int first_cnt;
    int cpu;                         /* >= 0: the logical CPU this thread pins itself to (one of the process' affinity mask) */
    int repeat;                      /* the thread's private copy holds its range `repeat` times over: the timed scans stream that much memory */
    int pinned;                      /* out: pthread_setaffinity_np succeeded */
} orc_thread_arg;

static void *orc_thread_main(void *p) {
    orc_thread_arg *a = (orc_thread_arg *)p;
    int64_t ids[256];
    double dist[256];
    /* the thread's range in memory it touched first itself: on a multi-socket host the pages then live on the thread's own NUMA node (one
     * numpy array, written by one thread, sits on one node: 256 threads streamed it at 84 GB/s, 2.1x one core) */
    a->pinned = 0;
    if (a->cpu >= 0) {                                               /* before the first touch: the pages follow the thread's CPU */
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(a->cpu, &set);
        a->pinned = pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0;
    }
    const size_t bytes = (size_t)a->n_rows * (size_t)a->row_stride;
    const int rep = a->repeat > 1 ? a->repeat : 1;
    uint8_t *mine = (uint8_t *)malloc(bytes * (size_t)rep);
    int have = 0;
    if (mine) { for (have = 0; have < rep; ++have) memcpy(mine + (size_t)have * bytes, a->rows, bytes); a->rows = mine; }
    else have = 1;
    /* the checked list: query 0 over the range itself (the first of the copies) */
    a->first_cnt = orc_scan_topk_with_fn(a->fn, a->queries, a->rows, a->n_rows, a->row_stride, a->dim, NULL, a->k, a->first_ids, a->first_dist);
    const int64_t timed_rows = a->n_rows * have;
    orc_scan_topk_with_fn(a->fn, a->queries, a->rows, timed_rows, a->row_stride, a->dim, NULL, a->k, ids, dist);       /* warm */
    __sync_fetch_and_add(a->ready, 1);
    while (!*a->go) sched_yield();
    int qi = 0;
    while (!*a->stop) {
        qi = qi + 1 == a->nq ? 0 : qi + 1;
        orc_scan_topk_with_fn(a->fn, a->queries + (int64_t)qi * a->query_stride, a->rows, timed_rows, a->row_stride, a->dim, NULL, a->k, ids, dist);
        a->scans += have;                                            /* in units of one pass over the thread's range */
    }
    free(mine);
    return NULL;
}

/* returns 0; out_first_*: nthreads x k (the per-range lists of query 0), out_counts: nthreads, out_rows_per_thread, out_elapsed (s),
 * out_scans: total completed range scans while the clock ran (vectors scanned = out_scans * rows_per_thread) */
int orc_scan_topk_threads(orc_distance_fn fn, const void *queries, int nq, int64_t query_stride, const void *rows, int64_t n_rows,
                          int64_t row_stride, int dim, int k, int nthreads, double seconds, int64_t *out_first_ids, double *out_first_dist,
                          int *out_counts, int64_t *out_rows_per_thread, double *out_elapsed, int64_t *out_scans) {
    return orc_scan_topk_threads_pinned(fn, queries, nq, query_stride, rows, n_rows, row_stride, dim, k, nthreads, seconds, NULL, 1, out_first_ids,
                                        out_first_dist, out_counts, out_rows_per_thread, out_elapsed, out_scans, NULL);
}

/* the same with thread i pinned to logical CPU cpus[i] (NULL: the scheduler places them) and every thread's private copy holding its range
 * `repeat` times over - a thread then streams repeat x P rows per timed scan (far beyond its caches) from memory its own CPU touched first;
 * out_pinned (may be NULL): threads whose pthread_setaffinity_np succeeded.  out_scans counts passes over P rows. */
int orc_scan_topk_threads_pinned(orc_distance_fn fn, const void *queries, int nq, int64_t query_stride, const void *rows, int64_t n_rows,
                                 int64_t row_stride, int dim, int k, int nthreads, double seconds, const int *cpus, int repeat,
                                 int64_t *out_first_ids, double *out_first_dist, int *out_counts, int64_t *out_rows_per_thread,
                                 double *out_elapsed, int64_t *out_scans, int *out_pinned) {
    if (nthreads < 1 || k < 1 || k > 256 || nq < 1 || n_rows < nthreads) return -1;
    const int64_t per = n_rows / nthreads;
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    orc_thread_arg *args = (orc_thread_arg *)calloc((size_t)nthreads, sizeof(orc_thread_arg));
    if (!th || !args) { free(th); free(args); return -2; }
    volatile int go = 0, stop = 0, ready = 0;
    int started = 0;
    for (int i = 0; i < nthreads; ++i) {
        orc_thread_arg *a = &args[i];
        a->fn = fn; a->queries = (const uint8_t *)queries; a->nq = nq; a->query_stride = query_stride;
        a->rows = (const uint8_t *)rows + (int64_t)i * per * row_stride; a->n_rows = per; a->row_stride = row_stride;
        a->dim = dim; a->k = k; a->go = &go; a->stop = &stop; a->ready = &ready; a->scans = 0;
        a->first_ids = out_first_ids + (int64_t)i * k; a->first_dist = out_first_dist + (int64_t)i * k;
        a->cpu = cpus ? cpus[i] : -1; a->repeat = repeat; a->pinned = 0;
        if (pthread_create(&th[i], NULL, orc_thread_main, a) != 0) break;
        ++started;
    }
    struct timespec t0, t1, nap;
    nap.tv_sec = 0; nap.tv_nsec = 2 * 1000 * 1000;                  /* every thread has copied its range and finished its first (warm) scan before the clock starts */
    for (int spin = 0; spin < 30000 && ready < started; ++spin) nanosleep(&nap, NULL);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    go = 1;
    nap.tv_sec = (time_t)seconds; nap.tv_nsec = (long)((seconds - (double)(time_t)seconds) * 1e9);
    nanosleep(&nap, NULL);
    stop = 1;
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    int64_t scans = 0;
    int pinned = 0;
    for (int i = 0; i < started; ++i) { scans += args[i].scans; out_counts[i] = args[i].first_cnt; pinned += args[i].pinned; }
    if (out_pinned) *out_pinned = pinned;
    *out_rows_per_thread = per;
    *out_elapsed = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    *out_scans = scans;
    free(th); free(args);
    return started == nthreads ? 0 : -3;
}

/* ------------------------------------------------------------------ quantizer */

/* (int)float as x86 cvttss2si does it: out-of-range / NaN -> INT_MIN (the f32 path at
 * sqlite-vector.c:524-527 has no guard, so this is what the reference binary computes on x86-64). */
static inline int trunc_i32(float r) {
    if (!(r > -2147483904.0f && r < 2147483648.0f)) return (int)0x80000000u;
    return (int)r;
}

/* sqlite-vector.c:495-504 */
static inline uint8_t round_u8(float s) {
    if (!isfinite(s)) return (s > 0.0f) ? 255u : 0u;
    float r = s + 0.5f * (1.0f - 2.0f * (s < 0.0f));
    if (r >= 255.0f) return 255u;
    if (r <= 0.0f) return 0u;
    return (uint8_t)(int)r;
}

/* sqlite-vector.c:506-515 */
static inline int8_t round_s8(float s) {
    if (!isfinite(s)) return (s > 0.0f) ? 127 : (s < 0.0f ? -128 : 0);
    float r = s + 0.5f * (1.0f - 2.0f * (s < 0.0f));
    if (r >= 127.0f) return 127;
    if (r <= -128.0f) return -128;
    return (int8_t)(int)r;
}

static inline float load_as_f32(int type, const void *src, int i) {
    switch (type) {
        case ORC_TYPE_F32: return ((const float *)src)[i];
        case ORC_TYPE_F16: return orc_f16_to_f32(((const uint16_t *)src)[i]);
        case ORC_TYPE_BF16: return orc_bf16_to_f32(((const uint16_t *)src)[i]);
        case ORC_TYPE_U8: return (float)((const uint8_t *)src)[i];
        case ORC_TYPE_I8: return (float)((const int8_t *)src)[i];
    }
    return 0.0f;
}

/* sqlite-vector.c:517-757 */
void orc_quantize(int type, const void *src, uint8_t *dst, float offset, float scale, int n, int qtype) {
    for (int i = 0; i < n; ++i) {
        float s = (load_as_f32(type, src, i) - offset) * scale;
        if (type == ORC_TYPE_F32) {
            /* f32 source: unguarded int conversion then integer clamp (:517-548, :626-656) */
            int r = trunc_i32(s + 0.5f * (1.0f - 2.0f * (s < 0.0f)));
            if (qtype == ORC_QUANT_U8) dst[i] = (uint8_t)(r > 255 ? 255 : (r < 0 ? 0 : r));
            else ((int8_t *)dst)[i] = (int8_t)(r > 127 ? 127 : (r < -128 ? -128 : r));
        } else {
            if (qtype == ORC_QUANT_U8) dst[i] = round_u8(s);
            else ((int8_t *)dst)[i] = round_s8(s);
        }
    }
}

/* sqlite-vector.c:1210-1268 */
void orc_quant_params(int type, const void *rows, int64_t n_rows, int64_t row_stride_bytes, int dim,
                      int *qtype_inout, float *scale, float *offset) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    int neg = 0;
    const uint8_t *p = (const uint8_t *)rows;
    for (int64_t r = 0; r < n_rows; ++r)
        for (int i = 0; i < dim; ++i) {
            float v = load_as_f32(type, p + r * row_stride_bytes, i);
            if (v < lo) lo = v;
            if (v > hi) hi = v;
            if (v < 0.0) neg = 1;
        }
    int qt = *qtype_inout;
    if (qt == 0) qt = neg ? ORC_QUANT_S8 : ORC_QUANT_U8;
    float amax = fmaxf(fabsf(lo), fabsf(hi));
    *scale = (qt == ORC_QUANT_U8) ? (255.0f / (hi - lo)) : (127.0f / amax);
    *offset = (qt == ORC_QUANT_U8) ? lo : 0.0f;
    *qtype_inout = qt;
}
