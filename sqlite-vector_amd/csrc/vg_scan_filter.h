// vg_scan_filter.h - single-query scan of an f32 corpus through its bf16 shadow copy (VG_SCAN_FILTER=0 turns it off).
//
// The f32 scan (vg_scan.h) is HBM-bound at ~85 % of the peak: the only way to answer faster is to read fewer bytes.
// This kernel streams the bf16 SHADOW copy of the corpus (half the bytes; built once per appended row for the batch
// path, vg_batch_h.hip) and computes s~ = sum q~ x~ with v_dot2c_f32_bf16.  bf16 keeps 8 bits of precision (7 stored):
// rounding to nearest has unit roundoff u = 2^-8, and BOTH the query and the row are rounded, so every product is off by
// a factor (1 + dq)(1 + dx), |dq|, |dx| <= u:  |s~ - s| <= (2u + u^2) sum |q_i x_i| <= 2^-7 (1 + 2^-9) |q||x|.  With the
// f32 summation error that gives |s~ - s| <= c |q||x|, c = 2^-7 (1 + 2^-9) + (D + 64) 2^-21, and with the cached ||x|| a
// LOWER bound of the row's distance.  (Round 1 used 2^-8 (1 + 2^-8) - one input's worth: rows whose elements all round
// the same way, e.g. a constant 1 + 2^-8 - 2^-20, could lose an exact duplicate of the query.  tests/test_gpu_filter_bound.py.)  A row whose bound cannot beat the wavefront's current k-th best is dropped; the others -
// k ln(N/k) per list plus a fraction of a row per query on random data - are re-evaluated by the whole wavefront on the
// f32 rows with the single-query kernel's accumulator (Accum<T_F32, ..>), and only that distance enters the list.  The
// answers are the f32 scan's answers bit for bit (the evaluation sums in vg_scan_kernel's order).  Rows the bound cannot judge (norm not finite or out of
// [1e-15, 1e15], i.e. Inf / NaN / huge / tiny rows) and queries with such a norm always take the exact path.
//
// Same decomposition as vg_scan_kernel: LPR lanes per row, U chunks (8 bf16 elements each) per lane, double-buffered
// loads, one sorted list per wavefront, vg_block_publish + vg_merge_kernel.  Metrics: L2, squared L2, dot.
#pragma once

#include "vg_scan.h"

struct FilterScanArgs {
    const uint8_t *shadow;     // N x bstride bytes of bf16 (zero padded)
    const uint8_t *rows;       // N x stride bytes of f32 (the corpus)
    const uint8_t *query;      // the f32 query, nch * 16 bytes, zero padded (device or pinned host)
    const float *row_norm;     // ||x|| per row
    uint64_t *cand;
    long long n_rows, stride, bstride;
    int nch, nch_b;            // 16-byte chunks per f32 / bf16 row
    int lpr_log2, k, root, dot, dim;
    float cerr;                // |s~ - s| <= cerr |q||x|: 2^-7 (1 + 2^-9) for the two rounded inputs + (D + 64) 2^-21 for the f32 sums
    float rel;                 // (D + 64) 2^-22: what the cached norms and the exact f32 evaluation themselves may be off by
    int xlpr_log2, xU;         // the launch shape vg_scan_kernel would use for this corpus: the exact evaluation sums in ITS order
    const uint64_t *init_keys; // the k best of a plain f32 scan over the first rows (64 keys) or nullptr: its k-th distance is
                               //   an upper bound of the final k-th best - the lists do not have to warm up from +Inf
    unsigned long long *evals; // instrumentation: += exact evaluations of this launch (one atomic per workgroup)
};

typedef __bf16 vgf_bf16x2 __attribute__((ext_vector_type(2)));

__device__ inline uint32_t vgf_pack_bf16(uint32_t lo, uint32_t hi) {           // two f32 bit patterns -> two bf16, round to nearest even
    return ((lo + 0x7FFFu + ((lo >> 16) & 1u)) >> 16) | ((hi + 0x7FFFu + ((hi >> 16) & 1u)) & 0xFFFF0000u);
}

template <int U, bool NT>
__global__ __launch_bounds__(VG_BLOCK) void vg_scan_filter_kernel(FilterScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & (VG_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr_log2 = a.lpr_log2, lpr = 1 << lpr_log2, rpb = VG_WAVE >> lpr_log2;
    const int sub = lane & (lpr - 1), rib = lane >> lpr_log2;
    const int k = a.k;

    // the f32 query, staged once per workgroup: the exact path reads it, the bf16 chunks are built from it
    uint4 *qs = reinterpret_cast<uint4 *>(smem);
    for (int c = threadIdx.x; c < a.nch; c += VG_BLOCK) qs[c] = reinterpret_cast<const uint4 *>(a.query)[c];
    __syncthreads();
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cb = sub + u * lpr;                                     // bf16 chunk = f32 chunks 2cb, 2cb+1
        const uint4 f0 = (2 * cb < a.nch) ? qs[2 * cb] : make_uint4(0u, 0u, 0u, 0u);
        const uint4 f1 = (2 * cb + 1 < a.nch) ? qs[2 * cb + 1] : make_uint4(0u, 0u, 0u, 0u);
        q[u] = make_uint4(vgf_pack_bf16(f0.x, f0.y), vgf_pack_bf16(f0.z, f0.w), vgf_pack_bf16(f1.x, f1.y), vgf_pack_bf16(f1.z, f1.w));
    }
    float qq;                                                             // sum q^2 (f32), every lane
    {
        Accum<T_F32, A_DOT> t;
        t.init();
        for (int c = lane; c < a.nch; c += VG_WAVE) t.chunk(qs[c], qs[c]);
        qq = vg_group_sum((t.a0 + t.a1) + (t.a2 + t.a3), 6);
    }
    const float qn = sqrtf(qq);
    const bool q_ok = (qq >= 1.0e-30f && qq <= 1.0e30f);                  // else: every row takes the exact path

    uint64_t mine = VG_EMPTY_KEY, thr = VG_EMPTY_KEY;
    unsigned n_exact = 0;                                                 // exact evaluations of this wavefront
    auto gate_of = [&](float t) -> float {                                // the bound must stay below this to go on
        if (a.dot) return t + a.rel * fabsf(t) + 1e-30f;
        const float t2 = a.root ? t * t : t;
        return t2 * (1.0f + 2.0f * a.rel) + 1e-30f;
    };
    float gate_init = INFINITY;
    if (a.init_keys) {
        const uint64_t kk = a.init_keys[k - 1];
        if (kk != VG_EMPTY_KEY) gate_init = gate_of(vg_sortable_f32((uint32_t)(kk >> 32)));
    }
    float thr_gate = gate_init;
    auto refresh_gate = [&]() {
        const float t = (thr == VG_EMPTY_KEY) ? INFINITY : vg_sortable_f32((uint32_t)(thr >> 32));
        thr_gate = fminf(gate_of(t), gate_init);
    };
    // the exact distance of one row (wave-uniform): the single-query f32 kernel's arithmetic IN ITS ORDER - lane group of
    // 2^xlpr_log2 lanes, lane `xs` takes chunks xs + u * xlpr (u < xU), same butterfly - so the distance is bit for bit what
    // vg_scan_kernel computes for the row (every group of the wavefront computes the same value)
    const int xlpr = 1 << a.xlpr_log2, xs = lane & (xlpr - 1);
    auto exact = [&](uint32_t row_u) -> float {
        const uint4 *xp = reinterpret_cast<const uint4 *>(a.rows + (unsigned long long)row_u * (unsigned long long)a.stride);
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        float d;
        if (a.dot) {
            Accum<T_F32, A_DOT> acc;
            acc.init();
            for (int u = 0; u < a.xU; ++u) { const int c = xs + u * xlpr; if (c < a.nch) acc.chunk(qs[c], xp[c]); else acc.chunk(zero, zero); }
            typename Accum<T_F32, A_DOT>::QStat s; s.qq = 0.0f;
            d = acc.finish(s, a.xlpr_log2, 0);
        } else {
            Accum<T_F32, A_L2> acc;
            acc.init();
            for (int u = 0; u < a.xU; ++u) { const int c = xs + u * xlpr; if (c < a.nch) acc.chunk(qs[c], xp[c]); else acc.chunk(zero, zero); }
            typename Accum<T_F32, A_L2>::QStat s; s.qq = 0.0f;
            d = acc.finish(s, a.xlpr_log2, a.root);
        }
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, vg_clamp(d))));
    };

    const long long nbatch = (a.n_rows + rpb - 1) / rpb;
    const long long wstride = (long long)gridDim.x * VG_WAVES_PER_BLOCK;
    long long b = (long long)blockIdx.x * VG_WAVES_PER_BLOCK + wave;
    uint4 cur[U], nxt[U];
    float nrm_cur = 0.0f, nrm_nxt = 0.0f;
    auto load = [&](uint4 (&dst)[U], float &nrm, long long batch) {
        vg_load_batch<U, NT>(dst, a.shadow, batch * rpb + rib, (batch < nbatch) ? a.n_rows : 0, a.bstride, sub, lpr, a.nch_b);
        const long long r0 = batch * rpb + rib;
        nrm = (batch < nbatch && r0 < a.n_rows) ? a.row_norm[r0] : 0.0f;
    };
    load(cur, nrm_cur, b);
    while (b < nbatch) {
        const long long bn = b + wstride;
        load(nxt, nrm_nxt, bn);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q[u].x), __builtin_bit_cast(vgf_bf16x2, cur[u].x), s0, false);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q[u].y), __builtin_bit_cast(vgf_bf16x2, cur[u].y), s1, false);
            s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q[u].z), __builtin_bit_cast(vgf_bf16x2, cur[u].z), s0, false);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vgf_bf16x2, q[u].w), __builtin_bit_cast(vgf_bf16x2, cur[u].w), s1, false);
        }
        const float st = vg_group_sum(s0 + s1, lpr_log2);
        const long long row = b * rpb + rib;
        const float nrm = nrm_cur, E = a.cerr * qn * nrm;
        const bool judged = q_ok && (nrm >= 1.0e-15f && nrm <= 1.0e15f);
        // lower bound of the distance (squared for L2)
        const float lb = a.dot ? -(st + E) - a.rel * qn * nrm : (qq + nrm * nrm - 2.0f * (st + E) - a.rel * (qq + nrm * nrm));
        const bool cand = (sub == 0) && (row < a.n_rows) && (!judged || lb < thr_gate);
        unsigned long long m = __ballot(cand);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t row_u = (uint32_t)__builtin_amdgcn_readlane((int)row, src);
            const float de = exact(row_u);
            ++n_exact;
            const uint64_t key = vg_make_key(de, row_u);
            if (de < INFINITY && key < thr) {                             // NaN / +Inf never enter (sqlite-vector.c:2102)
                vg_list_insert(mine, thr, key, lane, k);
                refresh_gate();
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        nrm_cur = nrm_nxt;
        b = bn;
    }
    __syncthreads();                                   // everyone is done with the query staging area
    if (a.evals) {                                     // one counter update per workgroup, nobody waits for it
        unsigned *blk = reinterpret_cast<unsigned *>(smem + VG_PUBLISH_LDS_BYTES - sizeof(unsigned));   // tail of the selection scratch: free until the publish
        if (threadIdx.x == 0) *blk = 0u;
        __syncthreads();
        if (lane == 0 && n_exact) atomicAdd(blk, n_exact);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.evals, (unsigned long long)*blk);
        __syncthreads();
    }
    vg_block_publish(smem, mine, k, a.cand + (long long)blockIdx.x * VG_WAVE);
}
