// vg_scan_multi.h - several queries per pass over the corpus.
//
// The batched entry point's path for every shape the matrix-core kernels (vg_batch.hip / vg_batch_i8.hip) do not
// serve on f32 / uint8 / int8 corpora: L1, rows > 512 floats / 1024 bytes, 32 < k <= 64.  NQ queries share every row
// load of the HBM-bound scan (f16 / bf16 scans are bound by their f64 arithmetic and gain nothing - vg_multi.hip).
// The arithmetic of each (query, row) pair is the single-query kernel's (same Accum, same epilogue), so int8 / uint8 results are bit-identical to the single scans and f32 results differ at most by the
// summation order of another launch shape.  Top-k mode only, k <= 64.
//   a.query : NQ zero-padded queries back to back (nch * 16 bytes each)
//   a.cand  : [NQ][gridDim.x][64] candidate keys, merged per query by vg_merge_kernel (grid NQ)
// Register budget: NQ * U query chunks + 2 * U row chunks per lane -> NQ = 4 with U <= 3, NQ = 2 with U <= 6.
#pragma once

#include "vg_scan.h"

template <int VT, int ACC, int U, int NQ, bool NT>          // VT: T_F32 / T_U8 / T_I8
__global__ __launch_bounds__(VG_BLOCK) void vg_scan_multi_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & (VG_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int lpr_log2 = a.lpr_log2;
    const int lpr = 1 << lpr_log2;
    const int rpb = VG_WAVE >> lpr_log2;
    const int sub = lane & (lpr - 1);
    const int rib = lane >> lpr_log2;

    uint4 *qs = reinterpret_cast<uint4 *>(smem);                       // [NQ][nch]
    for (int c = threadIdx.x; c < NQ * a.nch; c += VG_BLOCK) qs[c] = reinterpret_cast<const uint4 *>(a.query)[c];
    __syncthreads();
    uint4 q[NQ][U];
    typename Accum<VT, ACC>::QStat qstat[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = sub + u * lpr;
            q[n][u] = (c < a.nch) ? qs[n * a.nch + c] : make_uint4(0u, 0u, 0u, 0u);
        }
        qstat[n] = Accum<VT, ACC>::template query_stat<U>(q[n], lpr_log2);
    }
    uint64_t mine[NQ], thr[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) { mine[n] = VG_EMPTY_KEY; thr[n] = VG_EMPTY_KEY; }
    const int k = a.k;

    const long long nbatch = (a.n_rows + rpb - 1) / rpb;
    const long long wstride = (long long)gridDim.x * VG_WAVES_PER_BLOCK;
    long long b = (long long)blockIdx.x * VG_WAVES_PER_BLOCK + wave;
    uint4 cur[U], nxt[U];
    vg_load_batch<U, NT>(cur, a.rows, b * rpb + rib, (b < nbatch) ? a.n_rows : 0, a.stride, sub, lpr, a.nch);
    while (b < nbatch) {
        const long long bn = b + wstride;
        vg_load_batch<U, NT>(nxt, a.rows, bn * rpb + rib, (bn < nbatch) ? a.n_rows : 0, a.stride, sub, lpr, a.nch);
        const long long row = b * rpb + rib;
        const bool owner = (sub == 0) && (row < a.n_rows);
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            Accum<VT, ACC> acc;
            acc.init();
#pragma unroll
            for (int u = 0; u < U; ++u) acc.chunk(q[n][u], cur[u]);
            const float d = vg_clamp(acc.finish(qstat[n], lpr_log2, a.root));
            vg_list_offer(vg_make_key(d, (uint32_t)row), owner && (d < INFINITY), mine[n], thr[n], lane, k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        b = bn;
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        __syncthreads();                                   // query staging area / the previous publish is done with LDS
        vg_block_publish(smem, mine[n], k, a.cand + ((long long)n * gridDim.x + blockIdx.x) * VG_WAVE);
    }
}
