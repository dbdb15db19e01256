// vg_scan.h - the brute-force scan kernel (single query): HBM-bound streaming read of the N x D corpus with a
// fused wavefront-level top-k.  One template, instantiated per (element type, accumulation kind, U).
//
// Work decomposition (CDNA4):
//   * a row is `nch` 16-byte chunks; LPR = 2^lpr_log2 lanes share a row, lane `sub` owns chunks sub + u*LPR
//     (u < U), so one wave-wide global_load_dwordx4 reads 64/LPR rows x (LPR*16 B) contiguous bytes per row:
//     full 128-byte lines, every byte of the corpus fetched exactly once;
//   * a wavefront processes "batches" of 64/LPR rows in a grid-stride loop and keeps the next batch's U loads in
//     flight while it reduces the current one (loads straight to VGPRs, no LDS round trip: nothing is reused);
//   * the query lives in VGPRs (U x uint4 per lane), staged through LDS once per workgroup;
//   * per batch: U chunk folds -> butterfly over the lane group -> scalar epilogue -> clamp -> 64-bit key ->
//     ballot against the wave's current k-th best; only the rare survivors touch the sorted list;
//   * at the end the 16 wave lists of a workgroup merge through LDS (binary tree) and ONE list per workgroup -
//     i.e. per CU - goes to HBM (grid x 64 keys); vg_merge_kernel reduces those to the final k.
#pragma once

#include "vg_accum.h"
#include "vg_lists.h"

// LDS needed at the end of a scan workgroup: 16 wave lists + the selection scratch
#define VG_PUBLISH_LDS_BYTES (VG_WAVES_PER_BLOCK * VG_WAVE * 8 + VG_SEL_SCRATCH_BYTES)

// every wavefront deposits its sorted list in LDS, then the whole workgroup selects the k best into `dst` (global)
__device__ inline void vg_block_publish(uint8_t *smem, uint64_t mine, int k, uint64_t *dst) {
    const int lane = threadIdx.x & (VG_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);
    lists[wave * VG_WAVE + lane] = (lane < k) ? mine : VG_EMPTY_KEY;
    __syncthreads();
    vg_select_lists(lists, VG_WAVES_PER_BLOCK, k, dst, smem + VG_WAVES_PER_BLOCK * VG_WAVE * 8);
}

typedef uint32_t vg_u32x4 __attribute__((ext_vector_type(4)));

// NT = true streams the corpus with non-temporal loads (global_load_dwordx4 ... nt): every byte is read exactly
// once per query, so keeping it out of L2 / Infinity Cache is worth +11% on a 15 GB corpus (6.1 -> 6.8 TB/s
// measured).  NT = false is used when the whole corpus fits the 256 MiB Infinity Cache and repeated queries hit it.
template <bool NT>
__device__ inline uint4 vg_load16(const uint8_t *p) {
    if (NT) {
        vg_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const vg_u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4 *>(p);
}

// 16 zero bytes in device memory: where a lane that has nothing to load (a row behind the last one, a chunk behind the row's last
// one) points its load.  The loads of a batch are UNCONDITIONAL - only the address is selected: a load under `if (valid)` becomes a
// branch around the instruction, the compiler's wait-count bookkeeping must then assume that none of the prefetch loads were
// issued, and the wait in front of the CURRENT batch's arithmetic turns into vmcnt(0) - the wavefront sat out the full latency
// of the prefetch it had just issued before touching data that had long arrived (f16 U = 6, bf16 / uint8 U = 3: every shape
// whose loop the compiler unrolls into two register sets).  VG_LOAD_PREDICATED=1: the earlier form, for A/B runs.
#ifndef VG_LOAD_PREDICATED
#define VG_LOAD_PREDICATED 0
#endif
static __device__ __attribute__((aligned(16))) uint32_t vg_zero_chunk[4] = {0u, 0u, 0u, 0u};      // (not const: a constant-address-space pointer would turn the selected loads into flat loads)

template <int U, bool NT>
__device__ inline void vg_load_batch(uint4 (&dst)[U], const uint8_t *rows, long long row, long long n_rows,
                                     long long stride, int sub, int lpr, int nch) {
    const uint8_t *p = rows + row * stride + (long long)sub * 16;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = sub + u * lpr;
        if (VG_LOAD_PREDICATED) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < n_rows && c < nch) v = vg_load16<NT>(p + (long long)u * lpr * 16);
            dst[u] = v;
        } else {
            const uint8_t *src = (row < n_rows && c < nch) ? p + (long long)u * lpr * 16 : reinterpret_cast<const uint8_t *>(vg_zero_chunk);
            dst[u] = vg_load16<NT>(src);
        }
    }
}

#ifndef VG_HALF_LAUNDER
#define VG_HALF_LAUNDER 1
#endif
#ifndef VG_BF16_HOIST_Q
#define VG_BF16_HOIST_Q 1
#endif
#define VG_STORE_FLOATS 1024          // store mode: distances parked in LDS per wavefront between bursts of stores

// EX = true: the variants tie_order = reference needs (vg_reforder.hip) - a start threshold from a pass over the rows in front
// (init_keys), the candidate stream (emit) and "top-k + store" (out_dist != nullptr with k > 0: the replay's prefix pass).  They are
// instantiations of their own (vg_scan_ex.hip) because the plain kernels sit AT the 128-VGPR / 106-SGPR limit of 16 wavefronts per
// CU: the few registers the extras take spilled f32 U = 8 and f16 U = 6, shapes the plain scans use.
template <int VT, int ACC, int U, bool NT, bool EX = false>
__global__ __launch_bounds__(VG_BLOCK) void vg_scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ VgListExtras ex;
    const int lane = threadIdx.x & (VG_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr_log2 = a.lpr_log2;
    const int lpr = 1 << lpr_log2;
    const int rpb = VG_WAVE >> lpr_log2;          // rows per batch
    const int sub = lane & (lpr - 1);
    const int rib = lane >> lpr_log2;             // row in batch

    // ---- query: global -> LDS (once per workgroup) -> VGPRs
    uint4 *qs = reinterpret_cast<uint4 *>(smem);
    for (int c = threadIdx.x; c < a.nch; c += VG_BLOCK) qs[c] = reinterpret_cast<const uint4 *>(a.query)[c];
    // ---- candidate list (top-k mode).  tie_order = reference (vg_reforder.hip): a start threshold from the pass over the rows in
    // front and the candidate stream - kept in LDS (`ex`), published by the same barrier as the query
    const int k = a.k;
    uint64_t mine = VG_EMPTY_KEY;
    uint64_t thr = VG_EMPTY_KEY;
    if constexpr (EX) {
        if (k > 0) thr = vg_list_extras_init(&ex, a.init_keys, k, a.emit, a.emit_cap);
        if (a.emit_reset && blockIdx.x == 0 && threadIdx.x == 0) *a.emit_reset = 0ull;
    }
    __syncthreads();
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = sub + u * lpr;
        q[u] = (c < a.nch) ? qs[c] : make_uint4(0u, 0u, 0u, 0u);
    }
    const typename Accum<VT, ACC>::QStat qstat = Accum<VT, ACC>::template query_stat<U>(q, lpr_log2);
    const bool store_mode = EX ? ((a.out_dist != nullptr) && k == 0) : (a.out_dist != nullptr);
    const bool store_too = EX && (a.out_dist != nullptr) && k != 0; // the reference replay's prefix pass: top-k + every distance

    // ---- loop over row batches, one batch prefetched.  Top-k mode: grid-stride (batch b, b + W, ...).  Store mode:
    // each wavefront owns a CONTIGUOUS run of batches, parks VG_STORE_FLOATS distances in LDS and writes them out in
    // one burst of 16-byte-per-lane stores.  Stores share the load counter (vmcnt) and retire out of order with
    // loads, so the wait for the prefetched batch that follows a store drains EVERYTHING, prefetch included: with a
    // store every batch - or even every 16 batches - the kernel ran 10% slower than the top-k mode.  One burst per
    // 1024 rows makes that drain rare (measured: 2.46 -> 2.27 ms at 10M x 384; top-k mode 2.24).
    const long long nbatch = (a.n_rows + rpb - 1) / rpb;
    const long long nwaves = (long long)gridDim.x * VG_WAVES_PER_BLOCK;
    const long long gw = (long long)blockIdx.x * VG_WAVES_PER_BLOCK + wave;
    const int flush_every = VG_STORE_FLOATS / rpb;                    // iterations that fill the staging area
    long long per_wave = (nbatch + nwaves - 1) / nwaves;
    per_wave = ((per_wave + flush_every - 1) / flush_every) * flush_every;
    long long wstride = store_mode ? 1 : nwaves;
    long long b = store_mode ? gw * per_wave : gw;
    long long b_end = store_mode ? ((gw + 1) * per_wave < nbatch ? (gw + 1) * per_wave : nbatch) : nbatch;
    if (a.order == 1 && !store_mode) {                                // block-contiguous order (ScanArgs.order)
        const long long per_block = (nbatch + gridDim.x - 1) / gridDim.x;
        wstride = VG_WAVES_PER_BLOCK;
        b = (long long)blockIdx.x * per_block + wave;
        b_end = ((long long)blockIdx.x + 1) * per_block < nbatch ? ((long long)blockIdx.x + 1) * per_block : nbatch;
    }
    float *line = reinterpret_cast<float *>(smem + a.store_lds_off) + wave * VG_STORE_FLOATS;    // store mode only
    int in_line = 0;
    long long line_row0 = b * rpb;                                    // a multiple of VG_STORE_FLOATS: 16-byte aligned
    // Short uint8 / int8 rows (U <= 2) use a prefetch ring: NB buffers of U chunks, the loop unrolled NB times so that
    // every buffer keeps its registers (no copies); while buffer j is reduced the NB-1 others are in flight.  With
    // plain double buffering such rows have 1-2 KB per wavefront in flight and sit in s_waitcnt (dim-64 uint8 rows:
    // 4.0 -> 4.9 TB/s for L2, 4.8 -> 6.2 for dot).  Measured slower for f32 short rows and for U >= 3, which keep the
    // double buffer.
#ifndef VG_RING_F32
#define VG_RING_F32 0                 // > 0: f32 rows with 3 chunks per lane (the C2 / C4 shape, 40 VGPRs) run the ring with that many buffers
#endif
    constexpr bool RING_F32 = (VG_RING_F32 > 0) && VT == T_F32 && U == 3 && !EX;
    constexpr bool RING = ((U <= 2) && (VT == T_U8 || VT == T_I8)) || RING_F32;
    constexpr int NB = !RING ? 2 : RING_F32 ? (VG_RING_F32 > 2 ? VG_RING_F32 : 3) : (U == 2) ? 4 : 6;
    uint4 buf[NB][U];
    float nn[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) nn[j] = 0.0f;
    auto load = [&](uint4 (&dst)[U], float &nn_dst, long long batch) {
        vg_load_batch<U, NT>(dst, a.rows, batch * rpb + rib, (batch < b_end) ? a.n_rows : 0, a.stride, sub, lpr, a.nch);
        // A_COSN: the row's squared norm rides along with the batch prefetch (one dword per row from the cached vector)
        // (unconditional for the same reason as the chunk loads: row 0's norm stands in, the row's result is never used)
        if constexpr (ACC == A_COSN) { const long long r0 = batch * rpb + rib; nn_dst = a.row_nn[(batch < b_end && r0 < a.n_rows) ? r0 : 0]; }
    };
    auto process = [&](uint4 (&cur)[U], float nn_cur, long long bcur) {
        Accum<VT, ACC> acc;
        acc.init();
        if constexpr (VT == T_F16 || VT == T_BF16) {
            // keep the query as RAW halves in registers: without this the compiler hoists the widened (f32 / f64)
            // copies out of the loop - up to 48 more VGPRs per lane, which is what capped U (bytes in flight)
            // (bf16 with U <= 3 - its dot / cosine shapes - has the registers: there the hoisted f32 copies of the query save two
            // unpack operations per element pair, VG_BF16_HOIST_Q)
            if (VG_HALF_LAUNDER && !(VG_BF16_HOIST_Q && VT == T_BF16 && U <= 3)) {
#pragma unroll
                for (int u = 0; u < U; ++u) asm volatile("" : "+v"(q[u].x), "+v"(q[u].y), "+v"(q[u].z), "+v"(q[u].w));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc.chunk(q[u], cur[u]);
        const long long row = bcur * rpb + rib;
        const bool owner = (sub == 0) && (row < a.n_rows);
        float d;
        if constexpr (ACC == A_COSN) d = acc.finish_cached_norm(qstat, lpr_log2, nn_cur);
        else d = acc.finish(qstat, lpr_log2, a.root);
        if constexpr (VT == T_F16 || VT == T_BF16) {
            // rows (or a query) holding Inf/NaN: the owning lane replays the reference algorithm exactly (vg_half.h)
            if (acc.special(qstat, lpr_log2) && owner)
                d = vg_slow_distance<VT, (ACC == A_COSN ? A_COS : ACC)>(reinterpret_cast<const uint16_t *>(qs),
                                              reinterpret_cast<const uint16_t *>(a.rows + row * a.stride), a.dim, a.root);
        }
        d = vg_clamp(d);
        if (store_mode) {
            if (sub == 0) line[in_line * rpb + rib] = d;          // rows of consecutive batches are consecutive
            if (++in_line == flush_every) {
                // same wavefront wrote the lines: LDS ops are ordered, no barrier needed
                if (line_row0 + VG_STORE_FLOATS <= a.n_rows) {
#pragma unroll
                    for (int j = 0; j < VG_STORE_FLOATS / (4 * VG_WAVE); ++j)
                        reinterpret_cast<float4 *>(a.out_dist + line_row0)[j * VG_WAVE + lane] = reinterpret_cast<const float4 *>(line)[j * VG_WAVE + lane];
                } else {
                    for (int j = lane; j < VG_STORE_FLOATS && line_row0 + j < a.n_rows; j += VG_WAVE) a.out_dist[line_row0 + j] = line[j];
                }
                in_line = 0;
                line_row0 = (bcur + wstride) * rpb;
            }
        } else if constexpr (EX) {
            if (store_too && owner) a.out_dist[row] = d;
            vg_list_offer_ex(vg_make_key(d, (uint32_t)row), owner && (d < INFINITY), mine, thr, lane, k, &ex);
        } else {
            // NaN and +Inf never enter (strict '<' against INFINITY-initialised slots, sqlite-vector.c:1809,2102)
            vg_list_offer(vg_make_key(d, (uint32_t)row), owner && (d < INFINITY), mine, thr, lane, k);
        }
    };
    if constexpr (RING) {
#pragma unroll
        for (int j = 0; j < NB - 1; ++j) load(buf[j], nn[j], b + j * wstride);
        while (b < b_end) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const long long bj = b + j * wstride;             // the batch in buffer j
                load(buf[(j + NB - 1) % NB], nn[(j + NB - 1) % NB], bj + (NB - 1) * wstride);
                if (bj < b_end) process(buf[j], nn[j], bj);
            }
            b += NB * wstride;
        }
    } else {
        load(buf[0], nn[0], b);
        while (b < b_end) {
            const long long bn = b + wstride;
            load(buf[1], nn[1], bn);
            process(buf[0], nn[0], b);
#pragma unroll
            for (int u = 0; u < U; ++u) buf[0][u] = buf[1][u];
            nn[0] = nn[1];
            b = bn;
        }
    }
    if (store_mode) {
        // the run's last, partly filled staging area
        for (int j = lane; j < in_line * rpb && line_row0 + j < a.n_rows; j += VG_WAVE) a.out_dist[line_row0 + j] = line[j];
        return;
    }

    // ---- the workgroup's 16 wave lists -> ONE list per CU in HBM (parallel rank-select, vg_lists.h)
    __syncthreads();                                   // everyone is done with the query staging area
    vg_block_publish(smem, mine, k, a.cand + (long long)blockIdx.x * VG_WAVE);
}

// Long rows (more 16-byte chunks than 64 lanes x the largest register-resident U): one row per wavefront per step,
// the row is consumed in `S` slices of 64 x VG_LONG_U chunks with the accumulator carried across slices and the
// query slice read from LDS (ds_read_b128) instead of living in VGPRs.  Same epilogue / top-k tail as above.
#define VG_LONG_U 2
template <int VT, int ACC, bool NT>
__global__ __launch_bounds__(VG_BLOCK) void vg_scan_long_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & (VG_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    constexpr int U = VG_LONG_U;
    const int slice = VG_WAVE * U;                                  // chunks per slice
    const int S = (a.nch + slice - 1) / slice;
    const int nch_pad = S * slice;

    // ---- query: global -> LDS, zero padded to whole slices; the candidate lists reuse the space at the end
    uint4 *qs = reinterpret_cast<uint4 *>(smem);
    for (int c = threadIdx.x; c < nch_pad; c += VG_BLOCK)
        qs[c] = (c < a.nch) ? reinterpret_cast<const uint4 *>(a.query)[c] : make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // query statistics over the whole query (norm / special flags), folded slice by slice
    typename Accum<VT, ACC>::QStat qstat;
    {
        uint4 q0[U];
#pragma unroll
        for (int u = 0; u < U; ++u) q0[u] = qs[u * VG_WAVE + lane];
        qstat = Accum<VT, ACC>::template query_stat<U>(q0, 6);
        for (int s = 1; s < S; ++s) {
#pragma unroll
            for (int u = 0; u < U; ++u) q0[u] = qs[s * slice + u * VG_WAVE + lane];
            Accum<VT, ACC>::merge_qstat(qstat, Accum<VT, ACC>::template query_stat<U>(q0, 6));
        }
    }

    uint64_t mine = VG_EMPTY_KEY, thr = VG_EMPTY_KEY;
    const int k = a.k;
    const bool store_mode = (a.out_dist != nullptr);
    const long long wstride = (long long)gridDim.x * VG_WAVES_PER_BLOCK;
    const long long gw = (long long)blockIdx.x * VG_WAVES_PER_BLOCK + wave;

    // flattened (row, slice) stream per wavefront, one slice prefetched
    auto load_slice = [&](uint4 (&dst)[U], long long row, int s) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = s * slice + u * VG_WAVE + lane;
            if (VG_LOAD_PREDICATED) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (row < a.n_rows && c < a.nch) v = vg_load16<NT>(a.rows + row * a.stride + (long long)c * 16);
                dst[u] = v;
            } else {                                                // (unconditional, address selected: see vg_load_batch)
                dst[u] = vg_load16<NT>((row < a.n_rows && c < a.nch) ? a.rows + row * a.stride + (long long)c * 16
                                                                     : reinterpret_cast<const uint8_t *>(vg_zero_chunk));
            }
        }
    };
    uint4 cur[U], nxt[U];
    long long row = gw;
    load_slice(cur, row, 0);
    Accum<VT, ACC> acc;
    acc.init();
    int s = 0;
    while (row < a.n_rows) {
        long long nrow = row;
        int ns = s + 1;
        if (ns == S) { ns = 0; nrow = row + wstride; }
        load_slice(nxt, nrow, ns);
#pragma unroll
        for (int u = 0; u < U; ++u) acc.chunk(qs[s * slice + u * VG_WAVE + lane], cur[u]);
        if (ns == 0) {                                              // row complete
            float d = acc.finish(qstat, 6, a.root);
            if constexpr (VT == T_F16 || VT == T_BF16) {
                if (acc.special(qstat, 6) && lane == 0)
                    d = vg_slow_distance<VT, ACC>(reinterpret_cast<const uint16_t *>(qs),
                                                  reinterpret_cast<const uint16_t *>(a.rows + row * a.stride), a.dim, a.root);
            }
            d = vg_clamp(d);
            if (store_mode) {
                if (lane == 0) a.out_dist[row] = d;
            } else {
                vg_list_offer(vg_make_key(d, (uint32_t)row), (lane == 0) && (d < INFINITY), mine, thr, lane, k);
            }
            acc.init();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        row = nrow;
        s = ns;
    }
    if (store_mode) return;

    __syncthreads();                                   // everyone is done with the query staging area
    vg_block_publish(smem, mine, k, a.cand + (long long)blockIdx.x * VG_WAVE);
}


// (float) sum x^2 per row of an f16 / bf16 corpus, accumulated exactly like AccumHalf<.., A_COS> does it during a scan
// (f32 squares - exact for halves - widened to f64, f64 sums, one rounding to float at the end): the cosine scan then
// only has to accumulate the dot product (A_COSN).  16 lanes per row.  Rows holding Inf/NaN get whatever comes out:
// the scan sends them to the exact slow path anyway.
template <int VT>
__global__ __launch_bounds__(256) void vg_half_rownorm_kernel(const uint8_t *rows, long long row0, long long n, long long stride,
                                                              int nch, float *out) {
    const int sub = threadIdx.x & 15;
    const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ngroups = ((long long)gridDim.x * blockDim.x) >> 4;
    for (long long r = group; r < n; r += ngroups) {
        const uint4 *p = reinterpret_cast<const uint4 *>(rows + (row0 + r) * stride);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int c = sub; c < nch; c += 16) {
            const uint4 v = p[c];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lo, hi;
                vg_unpack2<VT>(w[j], lo, hi);
                if (j & 1) { s2 += (double)(lo * lo); s3 += (double)(hi * hi); }
                else { s0 += (double)(lo * lo); s1 += (double)(hi * hi); }
            }
        }
        const double s = vg_group_sum((s0 + s1) + (s2 + s3), 4);
        // (a sum that is not zero must not READ as zero: the filters judge a row of zeros - and only that - by its norm being exactly 0;
        //  bf16 elements below ~1e-23 square to less than the smallest float: such a row keeps the smallest one and stays unjudged)
        if (sub == 0) { const float f = (float)s; out[row0 + r] = (s > 0.0 && f == 0.0f) ? 1.401298464324817e-45f : f; }
    }
}
