// vg_scan_filter.h - single-query top-k scans through a cheap LOWER BOUND of every row's distance; only the rows whose
// bound can beat the wavefront's current k-th best are evaluated exactly (vg_scan_filter_kernel<XT, U, NT>).
//
// f32 corpora (XT = T_F32) - fewer BYTES.  The f32 scan (vg_scan.h) is HBM-bound at ~85 % of the peak: the only way to
// answer faster is to read fewer bytes.  The kernel streams the bf16 SHADOW copy of the corpus (half the bytes; built once
// per appended row, vg_batch_h.hip) and computes s~ = sum q~ x~ with v_dot2_f32_bf16.  bf16 keeps 8 bits of precision
// (7 stored): rounding to nearest has unit roundoff u = 2^-8, and BOTH the query and the row are rounded, so every product
// is off by a factor (1 + dq)(1 + dx), |dq|, |dx| <= u:  |s~ - s| <= (2u + u^2) sum |q_i x_i| <= 2^-7 (1 + 2^-9) |q||x|.
// With the f32 summation error: |s~ - s| <= c |q||x|, c = 2^-7 (1 + 2^-9) + (D + 64) 2^-21.  (Round 1 used
// 2^-8 (1 + 2^-8) - one input's worth: rows whose elements all round the same way, e.g. a constant 1 + 2^-8 - 2^-20, could
// lose an exact duplicate of the query.  tests/test_gpu_filter_bound.py.)
//
// Q8 = true - an int8 shadow copy: a QUARTER of an f32 corpus' bytes, HALF of an f16 / bf16 corpus'.  The shadow copy is int8: per row x = sx * xi + ex with integers
// xi in [-127, 127], sx = max|x| / 127 and ex the exact residual (vg_f32_to_q8_kernel stores sx and ||ex||, computed in f64
// and rounded up, next to the row); the query is split the same way in this kernel (q = sq * qi + eq).  Then
//     q.x = sq sx (qi.xi) + sq (qi.ex) + eq.x      |q.x - sq sx (qi.xi)| <= sq ||qi|| ||ex|| + ||eq|| ||x||   (Cauchy-Schwarz)
// with qi.xi an exact integer (v_dot4_i32_i8).  No rounding argument is involved: whatever integers the quantizer picked,
// the residuals are what is left.  The bound adapts to the data (||ex|| ~ sx sqrt(D / 12)): on N(0,1) rows of 384 floats
// ~5.6 against the bf16 copy's 3.0 - about twice the candidates for half the bytes.
//
// f16 / bf16 corpora (XT = T_F16 / T_BF16) - less ARITHMETIC.  Their plain scans follow the reference's f64 accumulation
// (distance-avx2.c:166-582) and that chain, not HBM, bounds them (5.2-6.4 TB/s).  Here the kernel reads the rows
// themselves (no shadow copy) and forms s~ in f32 - exact products of two halves, f32 sums: |s~ - s| <= c |q||x| with
// c = (D + 64) 2^-21 - which is a handful of VALU operations per 16 bytes; the f64 chain runs for the candidates only.
//
// With the cached per-row norm (f32 corpora: ||x||, f16 / bf16: (float) sum x^2) the bound of the distance is
//     L2 (squared)   |q|^2 + |x|^2 - 2 (s~ + c |q||x|)
//     dot            -(s~ + c |q||x|)
//     cosine         1 - (s~ + c |q||x|) / (|q||x|)
// each widened by what the norms / the float epilogue may be off by.  A row whose bound cannot beat the wavefront's
// current k-th best is dropped; the others - k ln(N/k) per list plus a fraction of a row per query on random data - are
// re-evaluated by the whole wavefront with the single-query kernel's own accumulator (Accum<XT, ..>) IN ITS summation
// order (same lanes-per-row x chunks-per-lane shape, same butterfly; f16 / bf16 rows holding Inf / NaN through the same
// slow path), and only that distance enters the list: the answers are the plain scan's answers bit for bit.  Rows the
// bound cannot judge (norm not finite or out of range, i.e. Inf / NaN / huge / tiny / zero rows) and queries with such a
// norm always take the exact path.
//
// Same decomposition as vg_scan_kernel: LPR lanes per row, U 16-byte chunks per lane, double-buffered loads, one sorted
// list per wavefront, vg_block_publish + vg_merge_kernel.
#pragma once

#include "vg_scan.h"

struct FilterScanArgs {
    const uint8_t *shadow;     // what the filter streams: N x bstride bytes (f32 corpora: the bf16 shadow copy; f16 / bf16: the rows)
    const uint8_t *rows;       // N x stride bytes: what the exact evaluation reads (the corpus)
    const uint8_t *query;      // the query in the corpus' element type, nch * 16 bytes, zero padded (device or pinned host)
    const float *row_norm;     // per row: ||x|| (f32 corpora) / (float) sum x^2 (f16 / bf16 corpora)
    const float2 *q8stat;      // int8 shadow copy only: per row (sx, ||ex|| rounded up); sx = NaN for rows the filter must not judge
    uint64_t *cand;
    long long n_rows, stride, bstride;
    int nch, nch_b;            // 16-byte chunks per corpus row / per streamed row
    int lpr_log2, k, root, mode, dim;   // mode: VGF_L2 (root: 1 = L2, 0 = squared), VGF_DOT, VGF_COS
    float cerr;                // |s~ - s| <= cerr |q||x|: (D + 64) 2^-21 for the f32 sums [+ 2^-7 (1 + 2^-9) for two bf16-rounded inputs]
    float rel;                 // (D + 64) 2^-22: what the cached norms and the exact evaluation themselves may be off by
    int xlpr_log2, xU;         // the launch shape vg_scan_kernel would use for this corpus: the exact evaluation sums in ITS order
    const uint64_t *init_keys; // the k best of a plain scan over the first rows (64 keys) or nullptr: its k-th distance is
                               //   an upper bound of the final k-th best - the lists do not have to warm up from +Inf
    const uint64_t *init_lists; // ... or that scan's per-CU candidate lists BEFORE their merge (n_init_lists x 64 keys): every workgroup takes
    int n_init_lists;           //   the k-th smallest list head itself (vg_kth_head) - one launch less in front of this kernel
    unsigned long long *evals; // instrumentation: += exact evaluations of this launch (one atomic per workgroup)
    unsigned long long *emit;  // tie_order = reference (vg_reforder.hip): [count | emit_cap pairs] - every row a list accepts whose distance
    unsigned emit_cap;         //   is strictly below init_keys' k-th distance, as (position << 32 | float bits); nullptr = off
};
enum { VGF_L2 = 0, VGF_DOT = 1, VGF_COS = 2, VGF_L1 = 3 /* f16 / bf16 only */ };

typedef __bf16 vgf_bf16x2 __attribute__((ext_vector_type(2)));

__device__ inline uint32_t vgf_pack_bf16(uint32_t lo, uint32_t hi) {           // two f32 bit patterns -> two bf16, round to nearest even
    return ((lo + 0x7FFFu + ((lo >> 16) & 1u)) >> 16) | ((hi + 0x7FFFu + ((hi >> 16) & 1u)) & 0xFFFF0000u);
}

// s += the 8 products of one 16-byte chunk of packed 16-bit elements (f32 sums of exact products)
template <int FT>
__device__ inline void vgf_dot_chunk(const uint4 &q, const uint4 &x, float &s0, float &s1) {
    if constexpr (FT == T_BF16) {
        s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q.x), __builtin_bit_cast(vgf_bf16x2, x.x), s0, false);
        s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q.y), __builtin_bit_cast(vgf_bf16x2, x.y), s1, false);
        s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q.z), __builtin_bit_cast(vgf_bf16x2, x.z), s0, false);
        s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q.w), __builtin_bit_cast(vgf_bf16x2, x.w), s1, false);
    } else {
        // f16: v_dot2_f32_f16.  The instruction may treat subnormal halves as zero: the bound carries a term for that (esub).
        s0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(vg_half2, q.x), __builtin_bit_cast(vg_half2, x.x), s0, false);
        s1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(vg_half2, q.y), __builtin_bit_cast(vg_half2, x.y), s1, false);
        s0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(vg_half2, q.z), __builtin_bit_cast(vg_half2, x.z), s0, false);
        s1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(vg_half2, q.w), __builtin_bit_cast(vg_half2, x.w), s1, false);
    }
}

// the 16 products of one 16-byte chunk of int8 elements, exact
__device__ inline void vgf_dot_chunk_i8(const uint4 &q, const uint4 &x, int &i0, int &i1) {
    i0 = __builtin_amdgcn_sdot4((int)q.x, (int)x.x, i0, false);
    i1 = __builtin_amdgcn_sdot4((int)q.y, (int)x.y, i1, false);
    i0 = __builtin_amdgcn_sdot4((int)q.z, (int)x.z, i0, false);
    i1 = __builtin_amdgcn_sdot4((int)q.w, (int)x.w, i1, false);
}
// the int8 image of one f32 element under scale s (inv = 1 / s up to rounding; any integer in [-127, 127] is a valid choice -
// the residual is defined against whatever comes out of here)
__device__ inline int vgf_q8(float v, float inv) {
    const float t = rintf(v * inv);
    return (int)fminf(fmaxf(t, -127.0f), 127.0f);
}
// s += sum |q - x| over the 8 elements of one 16-byte chunk: the reference's own f32 differences (distance-avx2.c:222-279 f16;
// bf16 subtracts in f64, :434-489 - an f32 difference of two bf16 values is off by at most 2^-24 of itself), summed in f32
// instead of f64.  Every term is >= 0, so the f32 sum is within (D + 64) 2^-23 of the f64 one, relatively: a lower bound of
// the L1 distance without any cached norm.
template <int FT>
__device__ inline void vgf_l1_chunk(const uint4 &q, const uint4 &x, float &s0, float &s1) {
    const uint32_t qw[4] = {q.x, q.y, q.z, q.w}, xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float q0, q1, x0, x1;
        vg_unpack2<FT>(qw[j], q0, q1);
        vg_unpack2<FT>(xw[j], x0, x1);
        s0 += fabsf(q0 - x0);
        s1 += fabsf(q1 - x1);
    }
}

// MODE (VGF_*) is a template parameter: as a run-time switch every instantiation carried all the exact evaluations (three
// sets of f64 accumulators for the half types) and the streaming loop spilled beyond 3 chunks per lane.
template <int XT, int MODE, int U, bool NT, bool Q8 = false>
__global__ __launch_bounds__(VG_BLOCK) void vg_scan_filter_kernel(FilterScanArgs a) {
    constexpr bool XF32 = (XT == T_F32);
    constexpr bool L1M = (MODE == VGF_L1);
    constexpr int mode = MODE;
    static_assert(!(XF32 && L1M), "the f32 L1 scan has no filter variant");
    constexpr int FT = XF32 ? T_BF16 : XT;                                // element type the filter multiplies
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & (VG_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr_log2 = a.lpr_log2, lpr = 1 << lpr_log2, rpb = VG_WAVE >> lpr_log2;
    const int sub = lane & (lpr - 1), rib = lane >> lpr_log2;
    const int k = a.k;

    // the query, staged once per workgroup: the exact path reads it, the filter's chunks are built from it
    uint4 *qs = reinterpret_cast<uint4 *>(smem);
    for (int c = threadIdx.x; c < a.nch; c += VG_BLOCK) qs[c] = reinterpret_cast<const uint4 *>(a.query)[c];
    __syncthreads();
    // int8 shadow: the query's own split q = sq * qi + eq (sq = max|q| / 127), from the query widened to f32 (f32 corpora: the
    // staged query itself; f16 / bf16: an exact f32 copy staged behind it)
    float q8_sq = 0.0f, q8_sqi = 0.0f, q8_eqn = 0.0f;                     // sq | sq ||qi|| | ||eq||, the last two rounded up
    bool q8_ok = true;
    float q8_inv = 0.0f, q8_qq = 0.0f;
    const float *qf = reinterpret_cast<const float *>(smem);              // 16 * nch_b floats, zero padded
    if constexpr (Q8) {
        const int nqf = 16 * a.nch_b;
        if constexpr (!XF32) {
            float *wq = reinterpret_cast<float *>(smem + (size_t)a.nch * 16);
            for (int e2 = threadIdx.x; e2 < nqf / 2; e2 += VG_BLOCK) {
                float lo = 0.0f, hi = 0.0f;
                if (e2 < a.nch * 4) vg_unpack2<XT>(reinterpret_cast<const uint32_t *>(smem)[e2], lo, hi);
                wq[2 * e2] = lo; wq[2 * e2 + 1] = hi;
            }
            __syncthreads();
            qf = wq;
        }
        const int nq_have = XF32 ? a.nch * 4 : nqf;                       // floats actually staged (f32: the corpus stride may end before 16 * nch_b)
        float mx = 0.0f;
        uint32_t bad = 0;
        for (int e = lane; e < nq_have; e += VG_WAVE) { const float f = qf[e]; mx = fmaxf(mx, fabsf(f)); bad |= !(fabsf(f) <= 3.0e38f); }
        mx = fmaxf(mx, vg_dpp<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_ROW_HALF_MIRROR>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_ROW_MIRROR>(mx));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        q8_ok = (__ballot(bad != 0) == 0) && mx >= 1.0e-30f && mx <= 1.0e30f;
        q8_sq = q8_ok ? mx / 127.0f : 1.0f;
        q8_inv = 1.0f / q8_sq;
        uint32_t i2 = 0;
        float e2s = 0.0f, q2s = 0.0f;
        for (int e = lane; e < nq_have; e += VG_WAVE) {
            const float f = qf[e];
            const int qi = vgf_q8(f, q8_inv);
            const float r = fmaf(-q8_sq, (float)qi, f);                  // one rounding
            i2 += (uint32_t)(qi * qi);
            e2s = fmaf(r, r, e2s);
            q2s = fmaf(f, f, q2s);
        }
        i2 = vg_group_sum(i2, 6);
        e2s = vg_group_sum(e2s, 6);
        q8_qq = vg_group_sum(q2s, 6);
        q8_sqi = q8_sq * sqrtf((float)i2) * (1.0f + 1.0e-5f);
        q8_eqn = sqrtf(e2s) * (1.0f + 1.0e-4f);                          // (f32 sum of D squares, each off by 2^-23 of itself at most)
    }
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cb = sub + u * lpr;
        if constexpr (Q8) {                                               // int8 chunk cb = the query's elements 16cb .. 16cb+15
            const int have = XF32 ? a.nch * 4 : 16 * a.nch_b;
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[j] = 0u;
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    const int e = 16 * cb + 4 * j + b4;
                    if (cb < a.nch_b && e < have) w[j] |= (uint32_t)(vgf_q8(qf[e], q8_inv) & 255) << (8 * b4);
                }
            }
            q[u] = make_uint4(w[0], w[1], w[2], w[3]);
        } else if constexpr (XF32) {                                      // bf16 chunk cb = f32 chunks 2cb, 2cb+1, rounded
            const uint4 f0 = (2 * cb < a.nch) ? qs[2 * cb] : make_uint4(0u, 0u, 0u, 0u);
            const uint4 f1 = (2 * cb + 1 < a.nch) ? qs[2 * cb + 1] : make_uint4(0u, 0u, 0u, 0u);
            q[u] = make_uint4(vgf_pack_bf16(f0.x, f0.y), vgf_pack_bf16(f0.z, f0.w), vgf_pack_bf16(f1.x, f1.y), vgf_pack_bf16(f1.z, f1.w));
        } else {
            q[u] = (cb < a.nch) ? qs[cb] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    float qq;                                                             // sum q^2 (f32, for the bound only), every lane
    if constexpr (XF32) {
        Accum<T_F32, A_DOT> t;
        t.init();
        for (int c = lane; c < a.nch; c += VG_WAVE) t.chunk(qs[c], qs[c]);
        qq = vg_group_sum((t.a0 + t.a1) + (t.a2 + t.a3), 6);
    } else {
        float t0 = 0.0f, t1 = 0.0f;
        for (int c = lane; c < a.nch; c += VG_WAVE) { const uint4 v = qs[c]; vgf_dot_chunk<FT>(v, v, t0, t1); }
        qq = vg_group_sum(t0 + t1, 6);
    }
    if constexpr (Q8 && !XF32) qq = q8_qq;                                // (exact products of the widened halves, no flushed subnormals)
    const float qn = sqrtf(qq);
    const bool q_ok = (qq >= 1.0e-30f && qq <= 1.0e30f) && q8_ok;         // else: every row takes the exact path
    // f16 only: v_dot2_f32_f16 may flush subnormal halves (|v| < 2^-14) to zero.  What s~ can lose that way:
    //   rows' subnormal elements   sum |q_i| 2^-14 <= 2^-14 |q|_1                      (esub_q, the same for every row)
    //   the query's subnormals     sum 2^-14 |x_i| <= 2^-14 sqrt(D) |x|                 (esub_x * |x|, zero unless the query has any)
    float esub_q = 0.0f, esub_x = 0.0f;
    if constexpr (FT == T_F16 && !Q8) {
        float l1 = 0.0f;
        uint32_t has_sub = 0;
        for (int c = lane; c < a.nch; c += VG_WAVE) {
            const uint4 v = qs[c];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lo, hi;
                vg_unpack2<T_F16>(w[j], lo, hi);
                l1 += fabsf(lo) + fabsf(hi);
                has_sub |= ((w[j] & 0x7C00u) == 0u && (w[j] & 0x03FFu) != 0u) || ((w[j] & 0x7C000000u) == 0u && (w[j] & 0x03FF0000u) != 0u);
            }
        }
        l1 = vg_group_sum(l1, 6);
        esub_q = 6.2e-5f * l1 * (1.0f + 1e-4f);
        if (__ballot(has_sub != 0) != 0) esub_x = 6.2e-5f * sqrtf((float)a.dim + 8.0f);
    }

    // ---- the exact evaluation's query statistics, in vg_scan_kernel's own shape (2^xlpr_log2 lanes per row, lane `xs` owns
    // chunks xs + u * xlpr, u < xU): Accum<XT, ..>::query_stat's arithmetic with a run-time chunk count
    const int xlpr = 1 << a.xlpr_log2, xs = lane & (xlpr - 1);
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    float xq_f32 = 0.0f;                                                  // f32 cosine: sum q^2 over the lane group
    double xq_f64 = 0.0;                                                  // f16 / bf16 cosine: the same in f64
    uint32_t xq_special = 0;                                              // f16 / bf16: the query holds Inf / NaN
    if constexpr (XF32) {
        if (mode == VGF_COS) {
            Accum<T_F32, A_DOT> t;
            t.init();
            for (int u = 0; u < a.xU; ++u) { const int c = xs + u * xlpr; const uint4 v = (c < a.nch) ? qs[c] : zero4; t.chunk(v, v); }
            xq_f32 = vg_group_sum((t.a0 + t.a1) + (t.a2 + t.a3), a.xlpr_log2);
        }
    } else {
        uint32_t sp = 0;
        double t = 0.0;
        for (int u = 0; u < a.xU; ++u) {
            const int c = xs + u * xlpr;
            const uint4 v = (c < a.nch) ? qs[c] : zero4;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sp |= vg_special_pair<XT>(w[j]);
                float lo, hi;
                vg_unpack2<XT>(w[j], lo, hi);
                t += (double)(lo * lo); t += (double)(hi * hi);
            }
        }
        xq_special = vg_group_or(sp, a.xlpr_log2);
        if (mode == VGF_COS) xq_f64 = vg_group_sum(t, a.xlpr_log2);
    }

    uint64_t mine = VG_EMPTY_KEY, thr = VG_EMPTY_KEY;
    unsigned n_exact = 0;                                                 // exact evaluations of this wavefront
    auto gate_of = [&](float t) -> float {                                // the bound must stay below this to go on
        if (mode != VGF_L2) return t + a.rel * fabsf(t) + 1e-30f;
        const float t2 = a.root ? t * t : t;
        return t2 * (1.0f + 2.0f * a.rel) + 1e-30f;
    };
    float gate_init = INFINITY;
    uint64_t emit_below = VG_EMPTY_KEY;                                   // (no pass in front: every accepted row may enter the slots)
    if (a.init_lists || a.init_keys) {
        // (the tail of the publish area is free until the publish; the host launches this form only when the staged query ends below it)
        const uint64_t kk = a.init_lists ? vg_kth_head(a.init_lists, a.n_init_lists, k, smem + VG_PUBLISH_LDS_BYTES - VG_KTH_HEAD_SCRATCH_BYTES - 16)
                                         : a.init_keys[k - 1];
        if (kk != VG_EMPTY_KEY) { gate_init = gate_of(vg_sortable_f32((uint32_t)(kk >> 32))); emit_below = kk & 0xFFFFFFFF00000000ull; }
    }
    float thr_gate = gate_init;
    auto refresh_gate = [&]() {
        const float t = (thr == VG_EMPTY_KEY) ? INFINITY : vg_sortable_f32((uint32_t)(thr >> 32));
        thr_gate = fminf(gate_of(t), gate_init);
    };
    // the exact distance of one row PER LANE GROUP (row_u is uniform within the 2^xlpr_log2 lanes of a group): the single-query
    // kernel's arithmetic IN ITS ORDER, so the distance is bit for bit what vg_scan_kernel computes for the row.
    // (Inlined on purpose: as a real function its registers are added to the kernel's and the streaming loop spills.)
    auto exact_with = [&](auto acc, uint32_t row_u) -> float {
        typedef decltype(acc) A;
        const uint8_t *rp = a.rows + (unsigned long long)row_u * (unsigned long long)a.stride;
        const uint4 *xp = reinterpret_cast<const uint4 *>(rp);
        acc.init();
        for (int u = 0; u < a.xU; ++u) { const int c = xs + u * xlpr; if (c < a.nch) acc.chunk(qs[c], xp[c]); else acc.chunk(zero4, zero4); }
        typename A::QStat st;
        float d;
        if constexpr (XF32) {
            st.qq = xq_f32;
            d = acc.finish(st, a.xlpr_log2, a.root);
        } else {
            st.qq = xq_f64; st.qspecial = xq_special;
            if constexpr (MODE == VGF_COS) d = acc.finish_cached_norm(st, a.xlpr_log2, a.row_norm[row_u]);
            else d = acc.finish(st, a.xlpr_log2, a.root);
            // rows (or a query) holding Inf / NaN: one lane replays the reference algorithm exactly (vg_half.h)
            if (acc.special(st, a.xlpr_log2) && xs == 0) {               // (the group's first lane: the one the caller reads)
                const uint16_t *q16 = reinterpret_cast<const uint16_t *>(qs), *r16 = reinterpret_cast<const uint16_t *>(rp);
                constexpr int SLOW = L1M ? A_L1 : (MODE == VGF_L2 ? A_L2 : (MODE == VGF_DOT ? A_DOT : A_COS));
                d = vg_slow_distance<XT, SLOW>(q16, r16, a.dim, a.root);
            }
        }
        return vg_clamp(d);
    };
    auto exact = [&](uint32_t row_u) -> float {
        constexpr int XACC = L1M ? A_L1 : (MODE == VGF_L2 ? A_L2 : (MODE == VGF_DOT ? A_DOT : (XF32 ? A_COS : A_COSN)));
        return exact_with(Accum<XT, XACC>(), row_u);
    };

    const long long nbatch = (a.n_rows + rpb - 1) / rpb;
    const long long wstride = (long long)gridDim.x * VG_WAVES_PER_BLOCK;
    long long b = (long long)blockIdx.x * VG_WAVES_PER_BLOCK + wave;
    uint4 cur[U], nxt[U];
    float nrm_cur = 0.0f, nrm_nxt = 0.0f;
    float2 q8_cur = make_float2(0.0f, 0.0f), q8_nxt = make_float2(0.0f, 0.0f);
    auto load = [&](uint4 (&dst)[U], float &nrm, float2 &q8s, long long batch) {
        vg_load_batch<U, NT>(dst, a.shadow, batch * rpb + rib, (batch < nbatch) ? a.n_rows : 0, a.bstride, sub, lpr, a.nch_b);
        // (per-row values: loaded unconditionally from a clamped index, then zeroed - a load under a branch costs the prefetch its
        // overlap, see vg_load_batch)
        const long long r0 = batch * rpb + rib;
        const bool live = batch < nbatch && r0 < a.n_rows;
        const long long ri = live ? r0 : 0;
        const float nv = a.row_norm[ri];
        nrm = live ? nv : 0.0f;
        if constexpr (Q8) { const float2 qv = a.q8stat[ri]; q8s = live ? qv : make_float2(0.0f, 0.0f); }
    };
    load(cur, nrm_cur, q8_cur, b);
    while (b < nbatch) {
        const long long bn = b + wstride;
        load(nxt, nrm_nxt, q8_nxt, bn);
        float s0 = 0.0f, s1 = 0.0f;
        if constexpr (Q8) {
            int i0 = 0, i1 = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) vgf_dot_chunk_i8(q[u], cur[u], i0, i1);
            s0 = (float)(int)vg_group_sum((uint32_t)(i0 + i1), lpr_log2) * (q8_sq * q8_cur.x);     // sq sx (qi.xi)
        } else if constexpr (L1M) {
            // keep the query as RAW halves in registers (the compiler would hoist the widened copies out of the loop and spill)
#pragma unroll
            for (int u = 0; u < U; ++u) asm volatile("" : "+v"(q[u].x), "+v"(q[u].y), "+v"(q[u].z), "+v"(q[u].w));
#pragma unroll
            for (int u = 0; u < U; ++u) vgf_l1_chunk<FT>(q[u], cur[u], s0, s1);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) vgf_dot_chunk<FT>(q[u], cur[u], s0, s1);
        }
        const float st = Q8 ? s0 : vg_group_sum(s0 + s1, lpr_log2);
        const long long row = b * rpb + rib;
        // cached norm: ||x|| for f32 corpora, sum x^2 for f16 / bf16 corpora
        const float nn = XF32 ? nrm_cur * nrm_cur : nrm_cur;
        const float nrm = XF32 ? nrm_cur : sqrtf(nrm_cur);
        // int8 shadow: sq ||qi|| ||ex|| + ||eq|| ||x|| (the cached norm is within rel of ||x||) + the three roundings of st
        const float E = Q8 ? q8_sqi * q8_cur.y + q8_eqn * nrm * (1.0f + a.rel) + 4.0e-7f * fabsf(st)
                           : a.cerr * qn * nrm + esub_q + esub_x * nrm;
        // (L1 needs no norm: its bound is the f32 sum itself; a NaN / Inf / overflowing sum sends the row to the exact path)
        // (a row of ZEROS is judged like any other - its estimate and its error term are exactly 0; a corpus with empty vectors would
        // otherwise pay an exact evaluation per such row and query.  Its cosine distance is the reference's 1.0: see lb below)
        const bool zero_row = !L1M && (nn == 0.0f);
        const bool judged = L1M ? (xq_special == 0u && st < 3.0e38f)
                                : (q_ok && (zero_row || ((XF32 ? (nrm >= 1.0e-15f && nrm <= 1.0e15f) : (nn >= 1.0e-30f && nn <= 1.0e30f)) &&
                                                         (!Q8 || (q8_cur.x > 0.0f && q8_cur.x <= 3.0e38f)))));       // (sx = NaN: Inf / NaN elements)
        // lower bound of the distance (squared for L2)
        float lb;
        if (L1M) lb = st - 2.0f * a.rel * st;
        else if (mode == VGF_L2) lb = qq + nn - 2.0f * (st + E) - a.rel * (qq + nn);
        else if (mode == VGF_DOT) lb = -(st + E) - a.rel * qn * nrm;
        else { const float r = (st + E) / (qn * nrm); lb = zero_row ? 1.0f - 4.0e-6f : 1.0f - r - a.rel * fabsf(r) - 4.0e-6f; }   // (norms + the float epilogue; zero norm: 1.0, distance-cpu.c:74-110)
        const bool cand = (sub == 0) && (row < a.n_rows) && (!judged || lb < thr_gate);
        unsigned long long m = __ballot(cand);
        while (m) {
            // Up to 64 / xlpr candidates at once: the exact evaluation runs in the plain kernel's shape - xlpr lanes per row - so
            // the wavefront's other lane groups, which used to compute the SAME row redundantly, each take a candidate of their
            // own (group g the g-th).  A dependent row fetch per candidate was what made unselective data expensive.
            const int ngrp = VG_WAVE >> a.xlpr_log2, gid = lane >> a.xlpr_log2;
            int mysrc = __ffsll((long long)m) - 1, ntake = 0;
            for (; ntake < ngrp && m; ++ntake) {
                const int s1 = __ffsll((long long)m) - 1;
                m &= m - 1;
                if (gid == ntake) mysrc = s1;
            }
            const uint32_t row_g = (uint32_t)__shfl((int)(uint32_t)row, mysrc);
            const float d_g = exact(row_g);
            for (int i = 0; i < ntake; ++i) {
                const int leader = i << a.xlpr_log2;
                const float de = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d_g), leader));
                const uint32_t row_u = (uint32_t)__builtin_amdgcn_readlane((int)row_g, leader);
                ++n_exact;
                const uint64_t key = vg_make_key(de, row_u);
                if (de < INFINITY && key < thr) {    // NaN / +Inf never enter (sqlite-vector.c:2102)
                    vg_list_insert(mine, thr, key, lane, k);
                    refresh_gate();
                    if (a.emit && key < emit_below) vg_emit_pair(a.emit, a.emit_cap, key, lane);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        nrm_cur = nrm_nxt;
        q8_cur = q8_nxt;
        b = bn;
    }
    __syncthreads();                                   // everyone is done with the query staging area
    if (a.evals) {                                     // one counter update per workgroup, nobody waits for it
        unsigned *blk = reinterpret_cast<unsigned *>(smem + VG_PUBLISH_LDS_BYTES - sizeof(unsigned));   // tail of the selection scratch: free until the publish
        if (threadIdx.x == 0) *blk = 0u;
        __syncthreads();
        if (lane == 0 && n_exact) atomicAdd(blk, n_exact);
        __syncthreads();
        if (threadIdx.x == 0 && *blk) atomicAdd(a.evals, (unsigned long long)*blk);
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(a.evals + 2, 1ull);      // (launches finished: what the host's guard averages over)
        __syncthreads();
    }
    vg_block_publish(smem, mine, k, a.cand + (long long)blockIdx.x * VG_WAVE);
}
