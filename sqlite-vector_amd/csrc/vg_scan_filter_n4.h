// vg_scan_filter_n4.h - the lower-bound filter scan (vg_scan_filter.h) for QUANTIZED corpora (uint8 / int8): half the bytes.
//
// uint8 / int8 rows have no smaller float representation to filter with - but they have a high nibble.  Per element
// x = 16 h + l with h = x >> 4 (arithmetic for int8: -8 .. 7; 0 .. 15 for uint8) and l = x & 15 in [0, 15].  The shadow copy
// keeps the h nibbles only (two elements per byte), and per row three exact statistics of what it drops:
//     xx = sum x^2 (integer, the row's squared norm)      Ls = sum l (integer)      ||l'|| with l' = l - 7.5  (rounded up)
// For a query q with mean mq = sum q / D and centred part q' = q - mq:
//     q.x = 16 q.h + q.l                      16 q.h = A: an exact integer, two v_dot4 per 8 elements (see below)
//     q.l = sum (q' + mq)(l' + 7.5) = q'.l' + mq Ls              (the cross terms cancel: sum q' = 0)
//     |q'.l'| <= ||q'|| ||l'||                                   (Cauchy-Schwarz)
// so  A + mq Ls - ||q'|| ||l'||  <=  q.x  <=  A + mq Ls + ||q'|| ||l'||.  Centring both factors is what makes the bound useful:
// l is close to uniform on 0 .. 15 whatever the data (||l'|| ~ 4.6 sqrt(D)), and ||q'|| is the query's spread, not its
// magnitude - on uniform bytes at D = 768 the slack is ~0.9 standard deviations of q.x over the rows, on bytes quantized from
// Gaussian embeddings about the same.  With the exact xx the bounds of the three metrics follow as in vg_scan_filter.h.
// Every candidate is re-evaluated with the plain kernel's integer accumulator (AccumInt: exact sums, the reference's float
// epilogue), so rowids and distance BITS equal the plain scan's - integer sums do not depend on a summation order at all.
//
// Nibble unpacking costs nothing extra: a shadow dword w holds the h nibbles of elements 8g .. 8g+3 in its low nibbles and
// 8g+4 .. 8g+7 in its high nibbles; (w << 4) & 0xF0F0F0F0 and w & 0xF0F0F0F0 are then four BYTES each holding 16 h - as an
// unsigned byte for uint8 (0 .. 240), as a signed byte for int8 (-128 .. 112) - i.e. exactly x - l, and v_dot4 against the
// query's own dwords 2g, 2g+1 accumulates A = sum q (x - l).
#pragma once

#include "vg_scan_filter.h"

// per row: (sum x^2, sum l, bits of ||l'|| rounded up, 0)
struct VgN4Stat { uint32_t xx, ls; float ln; uint32_t pad; };

template <int XT, int MODE, int U, bool NT>
__global__ __launch_bounds__(VG_BLOCK) void vg_scan_filter_n4_kernel(FilterScanArgs a) {
    static_assert(XT == T_U8 || XT == T_I8, "nibble filter: quantized corpora");
    static_assert(MODE == VGF_L2 || MODE == VGF_DOT || MODE == VGF_COS, "no L1 bound");
    constexpr int mode = MODE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & (VG_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr_log2 = a.lpr_log2, lpr = 1 << lpr_log2, rpb = VG_WAVE >> lpr_log2;
    const int sub = lane & (lpr - 1), rib = lane >> lpr_log2;
    const int k = a.k;

    uint4 *qs = reinterpret_cast<uint4 *>(smem);
    for (int c = threadIdx.x; c < a.nch; c += VG_BLOCK) qs[c] = reinterpret_cast<const uint4 *>(a.query)[c];
    __syncthreads();
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    // shadow chunk cb (32 elements) pairs with the query's chunks 2cb, 2cb+1
    uint4 qa[U], qb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cb = sub + u * lpr;
        qa[u] = (2 * cb < a.nch) ? qs[2 * cb] : zero4;
        qb[u] = (2 * cb + 1 < a.nch) ? qs[2 * cb + 1] : zero4;
    }
    // query statistics (exact integers; the pad bytes of the staged query are zero)
    uint32_t q2 = 0, q1 = 0;
    for (int c = lane; c < a.nch; c += VG_WAVE) {
        const uint4 v = qs[c];
        q2 = vg_dot4<XT>(v.x, v.x, q2); q2 = vg_dot4<XT>(v.y, v.y, q2); q2 = vg_dot4<XT>(v.z, v.z, q2); q2 = vg_dot4<XT>(v.w, v.w, q2);
        q1 = vg_dot4<XT>(v.x, 0x01010101u, q1); q1 = vg_dot4<XT>(v.y, 0x01010101u, q1);
        q1 = vg_dot4<XT>(v.z, 0x01010101u, q1); q1 = vg_dot4<XT>(v.w, 0x01010101u, q1);
    }
    q2 = vg_group_sum(q2, 6);
    q1 = vg_group_sum(q1, 6);
    const double qsum = (XT == T_U8) ? (double)q1 : (double)(int32_t)q1;
    const double mq_d = qsum / (double)a.dim;
    const double qp2 = fmax((double)q2 - qsum * mq_d, 0.0);                // ||q'||^2 = sum q^2 - (sum q)^2 / D
    const float mq = (float)mq_d;
    const float qnp = (float)sqrt(qp2) * (1.0f + 1.0e-6f) + 1.0e-30f;     // ||q'||, rounded up
    const float qq = (float)q2;
    const float qn = sqrtf(qq);
    const bool q_ok = q2 != 0u;                                           // a zero query: cosine is 1.0 for every row - exact path

    const int xlpr = 1 << a.xlpr_log2, xs = lane & (xlpr - 1);
    uint64_t mine = VG_EMPTY_KEY, thr = VG_EMPTY_KEY;
    unsigned n_exact = 0;
    auto gate_of = [&](float t) -> float {
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
    // the exact distance of one row per lane group (row_u uniform within a group): the plain kernel's integer accumulator and float epilogue
    auto exact = [&](uint32_t row_u) -> float {
        constexpr int XACC = (MODE == VGF_L2) ? A_L2 : (MODE == VGF_DOT ? A_DOT : A_COS);
        const uint4 *xp = reinterpret_cast<const uint4 *>(a.rows + (unsigned long long)row_u * (unsigned long long)a.stride);
        Accum<XT, XACC> acc;
        acc.init();
        for (int u = 0; u < a.xU; ++u) { const int c = xs + u * xlpr; if (c < a.nch) acc.chunk(qs[c], xp[c]); }
        typename Accum<XT, XACC>::QStat st;
        st.qq = q2;
        return vg_clamp(acc.finish(st, a.xlpr_log2, a.root));
    };

    const VgN4Stat *stats = reinterpret_cast<const VgN4Stat *>(a.q8stat);
    const long long nbatch = (a.n_rows + rpb - 1) / rpb;
    const long long wstride = (long long)gridDim.x * VG_WAVES_PER_BLOCK;
    long long b = (long long)blockIdx.x * VG_WAVES_PER_BLOCK + wave;
    uint4 cur[U], nxt[U];
    uint4 st_cur = zero4, st_nxt = zero4;
    auto load = [&](uint4 (&dst)[U], uint4 &rs, long long batch) {
        vg_load_batch<U, NT>(dst, a.shadow, batch * rpb + rib, (batch < nbatch) ? a.n_rows : 0, a.bstride, sub, lpr, a.nch_b);
        const long long r0 = batch * rpb + rib;                       // (unconditional load, clamped index: see vg_load_batch)
        const bool live = batch < nbatch && r0 < a.n_rows;
        const uint4 rv = *reinterpret_cast<const uint4 *>(stats + (live ? r0 : 0));
        rs = live ? rv : zero4;
    };
    load(cur, st_cur, b);
    while (b < nbatch) {
        const long long bn = b + wstride;
        load(nxt, st_nxt, bn);
        uint32_t a0 = 0, a1 = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
            const uint32_t ql[4] = {qa[u].x, qa[u].z, qb[u].x, qb[u].z}, qh[4] = {qa[u].y, qa[u].w, qb[u].y, qb[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0 = vg_dot4<XT>(ql[j], (w[j] << 4) & 0xF0F0F0F0u, a0);
                a1 = vg_dot4<XT>(qh[j], w[j] & 0xF0F0F0F0u, a1);
            }
        }
        const uint32_t A = vg_group_sum(a0 + a1, lpr_log2);              // = sum q (x - l), exact (|A| < 2^31 for D <= 16384)
        const long long row = b * rpb + rib;
        const float Af = (float)(int32_t)A;
        const float lsf = (float)st_cur.y, ln = __uint_as_float(st_cur.z);
        const float st = Af + mq * lsf;                                   // the estimate of q.x
        // the Cauchy-Schwarz term + the float roundings of the estimate itself (three operations on ~|A| + |mq| Ls)
        const float cs = qnp * ln;
        const float E = cs + 4.0e-7f * (fabsf(Af) + fabsf(mq) * lsf + cs) + 1.0e-30f;
        const float nn = (float)st_cur.x;
        const float nrm = sqrtf(nn);
        const bool judged = q_ok;                        // (a row of zeros too: estimate and error term are 0; its cosine distance is the reference's 1.0)
        float lb;
        if (mode == VGF_L2) lb = qq + nn - 2.0f * (st + E) - a.rel * (qq + nn);
        else if (mode == VGF_DOT) lb = -(st + E) - a.rel * qn * nrm;
        else { const float r = (st + E) / (qn * nrm); lb = (st_cur.x == 0u) ? 1.0f - 4.0e-6f : 1.0f - r - a.rel * fabsf(r) - 4.0e-6f; }
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
        st_cur = st_nxt;
        b = bn;
    }
    __syncthreads();
    if (a.evals) {
        unsigned *blk = reinterpret_cast<unsigned *>(smem + VG_PUBLISH_LDS_BYTES - sizeof(unsigned));
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

// rows of a uint8 / int8 corpus -> their nibble shadow copy + VgN4Stat; 16 lanes per row, a lane packs groups of 8 elements.
// The pad bytes of a corpus row are zero (h = l = 0): they add nothing to any sum, and ||l'|| is taken over the dim real elements.
template <int XT>
__global__ __launch_bounds__(256) void vg_to_n4_kernel(const uint8_t *rows, long long row0, long long n, long long stride, int dim,
                                                       uint8_t *out, long long ostride, VgN4Stat *stat) {
    const int l16 = threadIdx.x & 15;
    const long long groups = ((long long)gridDim.x * blockDim.x) >> 4;
    const long long n_pad = ((n + 3) / 4) * 4;
    const int ngroups_in = (int)(stride / 8), ngroups_out = (int)(ostride / 4);
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < n_pad; i += groups) {
        const bool live = i < n;
        const long long r = row0 + (live ? i : n - 1);
        const uint8_t *src = rows + r * stride;
        uint32_t xx = 0, ls = 0, l2 = 0;
        for (int g = l16; g < ngroups_out; g += 16) {
            uint32_t w = 0;
            if (g < ngroups_in) {
                const uint2 v = *reinterpret_cast<const uint2 *>(src + 8 * g);
                const uint32_t lo = v.x, hi = v.y;                       // elements 8g .. 8g+3 | 8g+4 .. 8g+7
                w = ((lo >> 4) & 0x0F0F0F0Fu) | (hi & 0xF0F0F0F0u);
                xx = vg_dot4<XT>(lo, lo, xx); xx = vg_dot4<XT>(hi, hi, xx);
                const uint32_t la = lo & 0x0F0F0F0Fu, lb = hi & 0x0F0F0F0Fu;
                ls = __builtin_amdgcn_udot4(la, 0x01010101u, ls, false); ls = __builtin_amdgcn_udot4(lb, 0x01010101u, ls, false);
                l2 = __builtin_amdgcn_udot4(la, la, l2, false); l2 = __builtin_amdgcn_udot4(lb, lb, l2, false);
            }
            if (live) *reinterpret_cast<uint32_t *>(out + r * ostride + 4 * g) = w;
        }
        xx = vg_group_sum(xx, 4); ls = vg_group_sum(ls, 4); l2 = vg_group_sum(l2, 4);
        if (live && l16 == 0) {
            const double v = (double)l2 - 15.0 * (double)ls + 56.25 * (double)dim;      // sum (l - 7.5)^2 over the dim elements
            VgN4Stat s;
            s.xx = xx; s.ls = ls; s.ln = (float)sqrt(v > 0.0 ? v : 0.0) * (1.0f + 1.0e-6f) + 1.0e-30f; s.pad = 0u;
            stat[r] = s;
        }
    }
}
