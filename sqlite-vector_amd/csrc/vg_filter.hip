// vg_filter.hip - host side of the filter scans (vg_scan_filter.h): which scans they serve, kernel selection, the launch with
// its plain pre-pass.  A translation unit of its own: the kernels are instantiated per (corpus type, metric mode, chunks per
// lane, load policy) - ~110 of them - and compile next to vg_api.hip's ~300 scan kernels instead of after them.
#include "vg_internal.h"

#include "vg_scan.h"
#include "vg_scan_filter.h"
#include "vg_scan_filter_n4.h"

typedef VgShape Shape;

// ---- f32 through the bf16 shadow copy (vg_scan_filter.h): half the bytes per row, exact answers.
typedef void (*filter_fn_t)(FilterScanArgs);

// Is the filter scan switched on for this corpus?  vg_corpus_set_scan_filter (the extension's scan_filter= option) wins
// over the VG_SCAN_FILTER environment switch; default on.
static bool scan_filter_enabled(const vg_corpus *c) {
    if (c->filter_disabled) return false;
    if (c->scan_filter_mode >= 0) return c->scan_filter_mode != 0;
    return vg_sw(SW_VG_SCAN_FILTER, 1) != 0;
}
// (vg_batch_api.hip: f32 batches follow the same switch and the same size threshold - they read the shadow copy these scans make)
bool vg_scan_filter_policy(const vg_corpus *c) {
    return scan_filter_enabled(c) && c->n_rows * c->stride >= (long long)vg_sw(SW_VG_SCAN_FILTER_MIN_MB, 3072) * (1ll << 20);
}
// Would a single top-k scan of this corpus go through a filter scan right now?  (vg_batch_api.hip: a handful of queries are then
// cheaper as single scans than as one 128- / 256-query-wide matrix pass.)
bool vg_scan_filter_would_serve(const vg_corpus *c, int metric, int k);
// exact-evaluation counters on the device ([0] filter scans, [1] filtered batches) + their pinned host mirror
int vg_ensure_filter_counters(vg_corpus *c) {
    if (c->d_filter_evals) return VG_OK;
    // [0] filter scans' exact evaluations, [1] filtered batches', [2] filter-scan launches finished (bumped by the kernels)
    HIP_TRY(hipMalloc(&c->d_filter_evals, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->d_filter_evals, 0, 4 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipHostMalloc(&c->h_filter_evals, 4 * sizeof(unsigned long long)));
    c->h_filter_evals[0] = c->h_filter_evals[1] = c->h_filter_evals[2] = c->h_filter_evals[3] = 0;
    return VG_OK;
}
// Which scans the filter serves: f32 / f16 / bf16 corpora, L2 / squared L2 / dot / cosine (f16 / bf16 cosine with the cached
// row norms, A_COSN - what the plain scan uses unless VG_HALF_COSN=0), and L1 on f16 / bf16 corpora.
// Small corpora keep the plain scan: the filter scan pays a pre-pass launch plus the exact evaluations of the lists'
// warm-up (measured at f32 D = 384: 3M rows 0.41 vs 0.70 ms, 1M rows - no pre-pass - 0.43 vs 0.26 ms, 10k rows 52 vs 34 us);
// f32 corpora additionally pay the shadow copy (+50 % HBM): from 3 GB up; f16 / bf16 corpora (no copy): from 1 GB up.
// (Those figures are for VG_SCAN_FILTER_SHADOW=bf16 / rows; the default int8 shadow copy has its own rule below.)
static bool filter_uses_q8(const vg_corpus *c, int metric);
// uint8 / int8 corpora: the nibble filter (vg_scan_filter_n4.h) - half the bytes.  Its bound assumes sums below 2^31 (the plain
// kernel's arithmetic is modular like the reference's): rows of at most 16384 elements.  Sizes (D = 768, measured): see below.
static bool n4_explicit(const vg_corpus *c) { return c->scan_filter_mode == 1 || vg_sw(SW_VG_SCAN_FILTER_N4, -1) == 1; }
static bool scan_filter_serves_n4(const vg_corpus *c, int metric) {
    if (c->vtype != VG_TYPE_U8 && c->vtype != VG_TYPE_I8) return false;
    if (metric != VG_DIST_L2 && metric != VG_DIST_SQUARED_L2 && metric != VG_DIST_DOT && metric != VG_DIST_COSINE) return false;
    if (c->dim > 16384 || c->n4_disabled || !scan_filter_enabled(c)) return false;
    // A 4-bit residual is coarse, and by Cauchy-Schwarz its bound is sqrt(D) looser than its typical size.  Measured at
    // 10M x 768 (profiles/r3d): bytes quantized from clustered unit-norm embeddings - the neighbours are much closer than a
    // random pair - 0.61 ms against 1.13 (10k exact rows per query); independent random bytes (the synthetic C3 corpus) -
    // distances concentrate, the slack is ~3 standard deviations of them - most rows are candidates.  So the corpus is PROBED
    // before + 52 % device memory is spent on it (n4_probe, see vg_launch_scan_filter): the first eligible scan runs the filter
    // over a 2M-row prefix only and counts; a selective prefix switches the filter on, an unselective one leaves the corpus
    // with the plain kernel (probed again once it has doubled).  vg_corpus_set_scan_filter(c, 1) / the extension's
    // scan_filter=1 / VG_SCAN_FILTER_N4=1 switch it on without a probe (the guard still watches), VG_SCAN_FILTER_N4=0 off.
    if (!n4_explicit(c)) {
        if (vg_sw(SW_VG_SCAN_FILTER_N4, -1) == 0) return false;
        if (c->n4_probe == 2 && c->n_rows < 2 * c->n4_probe_rows) return false;
    }
    if (vg_sw(SW_VG_SCAN_FILTER_MIN_MB, -1) >= 0) return c->n_rows * c->stride >= (long long)vg_sw(SW_VG_SCAN_FILTER_MIN_MB, 0) * (1ll << 20);
    return c->n_rows >= (1 << 20) && c->n_rows * c->stride >= (768ll << 20);
}
static bool scan_filter_serves(const vg_corpus *c, int metric) {
    if (c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8) return scan_filter_serves_n4(c, metric);
    const bool half = (c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16);
    if (c->vtype != VG_TYPE_F32 && !half) return false;
    if (metric == VG_DIST_L1 && !half) return false;         // (the f32 L1 scan already streams at the HBM ceiling: nothing to skip)
    if (metric != VG_DIST_L2 && metric != VG_DIST_SQUARED_L2 && metric != VG_DIST_DOT && metric != VG_DIST_COSINE && metric != VG_DIST_L1) return false;
    if (half && metric == VG_DIST_COSINE && !vg_sw(SW_VG_HALF_COSN, 1)) return false;
    if (!scan_filter_enabled(c)) return false;
    // int8 shadow copy (measured, profiles/r3a_int8_filter_size_threshold.txt, D = 384): from 2^20 rows - where the pre-pass
    // starts - the filter wins at every size tried (1.2M rows: 0.09 + 0.035 ms against 0.27 f32 / 0.17 f16); below, without a
    // pre-pass, it only ties.  In bytes: a quarter / half of the stream has to buy back ~45 us of pre-pass and second launch.
    if (filter_uses_q8(c, metric) && vg_sw(SW_VG_SCAN_FILTER_MIN_MB, -1) < 0)
        return c->n_rows >= (1 << 20) && c->n_rows * c->stride >= (512ll << 20);
    return c->n_rows * c->stride >= (long long)vg_sw(SW_VG_SCAN_FILTER_MIN_MB, half ? 1024 : 3072) * (1ll << 20);
}

template <int XT, int MODE, bool NT, bool Q8>
static filter_fn_t pick_filter_u(int U) {
    switch (U) {
        case 1: return vg_scan_filter_kernel<XT, MODE, 1, NT, Q8>;
        case 2: return vg_scan_filter_kernel<XT, MODE, 2, NT, Q8>;
        case 3: return vg_scan_filter_kernel<XT, MODE, 3, NT, Q8>;
        case 4: return vg_scan_filter_kernel<XT, MODE, 4, NT, Q8>;
        case 6: return vg_scan_filter_kernel<XT, MODE, 6, NT, Q8>;
    }
    return nullptr;
}
template <int XT, bool NT, bool Q8>
static filter_fn_t pick_filter_mode(int mode, int U) {
    switch (mode) {
        case VGF_L2: return pick_filter_u<XT, VGF_L2, NT, Q8>(U);
        case VGF_DOT: return pick_filter_u<XT, VGF_DOT, NT, Q8>(U);
        case VGF_COS: return pick_filter_u<XT, VGF_COS, NT, Q8>(U);
        case VGF_L1:
            if constexpr (XT != T_F32) return pick_filter_u<XT, VGF_L1, NT, false>(U);
            return nullptr;
    }
    return nullptr;
}
template <bool NT>
static filter_fn_t pick_filter(int vtype, int mode, int U, bool q8 = false) {
    switch (vtype) {
        case VG_TYPE_F32: return q8 ? pick_filter_mode<T_F32, NT, true>(mode, U) : pick_filter_mode<T_F32, NT, false>(mode, U);
        case VG_TYPE_F16: return q8 ? pick_filter_mode<T_F16, NT, true>(mode, U) : pick_filter_mode<T_F16, NT, false>(mode, U);
        case VG_TYPE_BF16: return q8 ? pick_filter_mode<T_BF16, NT, true>(mode, U) : pick_filter_mode<T_BF16, NT, false>(mode, U);
    }
    return nullptr;
}

// ---- the int8 shadow copy of a corpus (vg_scan_filter.h, Q8): 16 lanes per row.  Row r becomes ostride bytes of int8
// (zero padded) and stat[r] = (sx, ||ex||): sx = max|x| / 127, ex = x - sx * xi formed and summed in f64 (the product of a
// float and a 7-bit integer is exact there; f16 / bf16 elements widen to f32 exactly), its norm rounded up.  Rows with
// Inf / NaN elements get sx = NaN (never judged).
template <int XT> __device__ inline float vgq_elem(const uint8_t *row, int e) {
    if constexpr (XT == T_F32) return reinterpret_cast<const float *>(row)[e];
    else {
        float lo, hi;
        vg_unpack2<XT>(reinterpret_cast<const uint32_t *>(row)[e >> 1], lo, hi);
        return (e & 1) ? hi : lo;
    }
}
template <int XT>
__global__ __launch_bounds__(256) void vg_to_q8_kernel(const uint8_t *rows, long long row0, long long n, long long stride, int dim,
                                                       uint8_t *out, long long ostride, float2 *stat) {
    const int l16 = threadIdx.x & 15;
    const long long groups = ((long long)gridDim.x * blockDim.x) >> 4;
    const long long n_pad = ((n + 3) / 4) * 4;                            // whole wavefronts stay together (DPP reductions)
    const int dim2 = (dim + 1) & ~1;                                      // (halves come in pairs; the pad element of an odd row is zero)
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < n_pad; i += groups) {
        const bool live = i < n;
        const long long r = row0 + (live ? i : n - 1);
        const uint8_t *src = rows + r * stride;
        float mx = 0.0f;
        uint32_t bad = 0;
        for (int e = 4 * l16; e < dim2; e += 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (e + j < dim) { const float v = vgq_elem<XT>(src, e + j); mx = fmaxf(mx, fabsf(v)); bad |= !(fabsf(v) <= 3.0e38f); }
        }
        mx = fmaxf(mx, vg_dpp<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_ROW_HALF_MIRROR>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_ROW_MIRROR>(mx));
        bad = vg_group_or(bad, 4);
        const float sx = (mx > 0.0f) ? mx / 127.0f : 0.0f;
        const float inv = (mx > 0.0f) ? 1.0f / sx : 0.0f;
        double e2 = 0.0;
        for (int e = 4 * l16; e < (int)ostride; e += 64) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (e + j < dim && !bad) {
                    const float v = vgq_elem<XT>(src, e + j);
                    const int xi = vgf_q8(v, inv);
                    const double res = (double)v - (double)sx * (double)xi;
                    e2 += res * res;
                    w |= (uint32_t)(xi & 255) << (8 * j);
                }
            }
            if (live) *reinterpret_cast<uint32_t *>(out + r * ostride + e) = w;
        }
        e2 = vg_group_sum(e2, 4);
        if (live && l16 == 0) {
            float ex = (float)sqrt(e2);
            ex = ex * (1.0f + 4.0e-7f) + 1.0e-37f;                        // rounded up (f64 sum, one sqrt, one conversion)
            stat[r] = bad ? make_float2(__builtin_nanf(""), 0.0f) : make_float2(sx, e2 > 0.0 ? ex : 0.0f);
        }
    }
}

// The same pass with the row held in REGISTERS between the two sweeps (max |x|, then quantize + residual): lane l of a row's 16 keeps
// the 16-byte chunks l, l + 16, ... (U of them), so every corpus byte is fetched once - the kernel above reads each row twice
// (rocprofv3 FETCH_SIZE 30.6 GB for a 15.36 GB corpus).  Rows of up to 16 U chunks (f32: 64 U elements, f16 / bf16: 128 U).
template <int XT, int U>
__global__ __launch_bounds__(256) void vg_to_q8_reg_kernel(const uint8_t *rows, long long row0, long long n, long long stride, int dim,
                                                           int nch, uint8_t *out, long long ostride, float2 *stat) {
    constexpr int N = (XT == T_F32) ? 4 : 8;                               // elements per 16-byte chunk
    const int l16 = threadIdx.x & 15;
    const long long groups = ((long long)gridDim.x * blockDim.x) >> 4;
    const long long n_pad = ((n + 3) / 4) * 4;                            // whole wavefronts stay together (DPP reductions)
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < n_pad; i += groups) {
        const bool live = i < n;
        const long long r = row0 + (live ? i : n - 1);
        const uint8_t *src = rows + r * stride;
        // all of the lane's chunks are requested before the first one is touched - unconditionally, a chunk behind the row's last one
        // reads 16 zero bytes (vg_load_batch's reasoning: a load under a branch is waited for on the spot, and this kernel's first
        // form had ONE load in flight per wavefront) - and the elements behind `dim` are zeroed by a select, not skipped by a branch
        uint4 raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = l16 + 16 * u;
            raw[u] = vg_load16<true>(c < nch ? src + (long long)c * 16 : reinterpret_cast<const uint8_t *>(vg_zero_chunk));
        }
        float v[U][N];
        float mx = 0.0f;
        uint32_t bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = l16 + 16 * u;
            const uint32_t w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (XT == T_F32) v[u][j] = __uint_as_float(w[j]);
                else vg_unpack2<XT>(w[j], v[u][2 * j], v[u][2 * j + 1]);
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                v[u][j] = (c * N + j < dim) ? v[u][j] : 0.0f;
                const float av = fabsf(v[u][j]);
                mx = fmaxf(mx, av);
                bad |= !(av <= 3.0e38f) ? 1u : 0u;
            }
        }
        mx = fmaxf(mx, vg_dpp<VG_DPP_QUAD_PERM(1, 0, 3, 2)>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_QUAD_PERM(2, 3, 0, 1)>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_ROW_HALF_MIRROR>(mx));
        mx = fmaxf(mx, vg_dpp<VG_DPP_ROW_MIRROR>(mx));
        bad = vg_group_or(bad, 4);
        const float sx = (mx > 0.0f) ? mx / 127.0f : 0.0f;
        const float inv = (mx > 0.0f) ? 1.0f / sx : 0.0f;
        // The residual norm in f32, in units of the row's scale (round 6; it was four f64-rate instructions per element - the pass ran at
        // 0.65-0.69 of the HBM peak, VALU-bound): res = fma(-sx, xi, x) is the exact residual rounded ONCE; rho = res / sx lies in
        // [-0.5, 0.5] whatever the row's magnitude (no underflow of its square); sum rho^2 over <= 512 elements in f32.  Relative error of
        // ||ex|| = sx sqrt(sum rho^2): < 2e-6 - the result is rounded up by 4e-6 (the f64 sum was rounded up by 4e-7).
        float e2 = 0.0f;
        uint8_t *o = out + r * ostride;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = l16 + 16 * u;
            uint32_t w[N / 4];
#pragma unroll
            for (int j4 = 0; j4 < N / 4; ++j4) {
                w[j4] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                       // (an element behind `dim` is 0 here: image 0, residual 0)
                    const float x = v[u][4 * j4 + j];
                    const float tq = fminf(fmaxf(rintf(x * inv), -127.0f), 127.0f);       // (vgf_q8, the integer kept as a float)
                    const int xi = (int)tq;
                    const float rho = fmaf(-sx, tq, x) * inv;
                    e2 = fmaf(rho, rho, e2);
                    w[j4] |= (uint32_t)(xi & 255) << (8 * j);
                }
                if (bad) w[j4] = 0;                                 // (a row the filter must not judge: zero image, NaN scale below)
            }
            if (live && c * N < (int)ostride) {
                if constexpr (N == 4) *reinterpret_cast<uint32_t *>(o + c * 4) = w[0];
                else *reinterpret_cast<uint2 *>(o + c * 8) = make_uint2(w[0], w[1]);
            }
        }
        // the shadow row's zero padding behind the chunks the corpus row has
        if (live) for (int e = N * nch + 4 * l16; e < (int)ostride; e += 64) *reinterpret_cast<uint32_t *>(o + e) = 0u;
        e2 = vg_group_sum(e2, 4);
        if (live && l16 == 0) {
            float ex = sx * sqrtf(e2);
            ex = ex * (1.0f + 4.0e-6f) + 1.0e-37f;                        // rounded up
            stat[r] = bad ? make_float2(__builtin_nanf(""), 0.0f) : make_float2(sx, e2 > 0.0f ? ex : 0.0f);
        }
    }
}
typedef void (*to_q8_fn_t)(const uint8_t *, long long, long long, long long, int, int, uint8_t *, long long, float2 *);
template <int XT> static to_q8_fn_t pick_to_q8_reg(int nch) {
    if (nch <= 32) return vg_to_q8_reg_kernel<XT, 2>;
    if (nch <= 64) return vg_to_q8_reg_kernel<XT, 4>;
    if (nch <= 96) return vg_to_q8_reg_kernel<XT, 6>;
    if (nch <= 128) return vg_to_q8_reg_kernel<XT, 8>;
    return nullptr;
}

template <int XT, int MODE, bool NT>
static filter_fn_t pick_n4_u(int U) {
    switch (U) {
        case 1: return vg_scan_filter_n4_kernel<XT, MODE, 1, NT>;
        case 2: return vg_scan_filter_n4_kernel<XT, MODE, 2, NT>;
        case 3: return vg_scan_filter_n4_kernel<XT, MODE, 3, NT>;
        case 4: return vg_scan_filter_n4_kernel<XT, MODE, 4, NT>;
        case 6: return vg_scan_filter_n4_kernel<XT, MODE, 6, NT>;
    }
    return nullptr;
}
template <bool NT>
static filter_fn_t pick_n4(int vtype, int mode, int U) {
    if (vtype == VG_TYPE_U8) return mode == VGF_L2 ? pick_n4_u<T_U8, VGF_L2, NT>(U) : (mode == VGF_DOT ? pick_n4_u<T_U8, VGF_DOT, NT>(U) : pick_n4_u<T_U8, VGF_COS, NT>(U));
    return mode == VGF_L2 ? pick_n4_u<T_I8, VGF_L2, NT>(U) : (mode == VGF_DOT ? pick_n4_u<T_I8, VGF_DOT, NT>(U) : pick_n4_u<T_I8, VGF_COS, NT>(U));
}
static long long n4_shadow_stride(const vg_corpus *c) { return (((long long)c->dim + 31) / 32) * 16; }
int vg_ensure_n4_shadow(vg_corpus *c, int64_t upto_rows) {
    const long long ns = n4_shadow_stride(c);
    upto_rows = std::min<int64_t>(upto_rows, c->n_rows);
    if (c->n4_cap < upto_rows) {
        // (a probe asks for a prefix only: it gets a prefix-sized allocation, not the corpus' + 52 %)
        const int64_t cap = (upto_rows < c->n_rows) ? upto_rows : std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rows_n4) hipFree(c->d_rows_n4);
        if (c->d_n4stat) hipFree(c->d_n4stat);
        c->d_rows_n4 = nullptr; c->d_n4stat = nullptr; c->n4_cap = 0; c->n4_rows = 0;
        // both or neither: a failed second allocation must not leave the first one pinned while the corpus falls back to the plain scan
        if (hipMalloc(&c->d_rows_n4, (size_t)cap * ns) != hipSuccess || hipMalloc(&c->d_n4stat, (size_t)cap * sizeof(VgN4Stat)) != hipSuccess) {
            (void)hipGetLastError();
            if (c->d_rows_n4) hipFree(c->d_rows_n4);
            if (c->d_n4stat) hipFree(c->d_n4stat);
            c->d_rows_n4 = nullptr; c->d_n4stat = nullptr;
            return vg_fail(VG_ERR_NOMEM, "no device memory for the high-nibble shadow copy (%lld rows)", (long long)cap);
        }
        c->n4_cap = cap;
    }
    upto_rows = std::min<int64_t>(upto_rows, c->n_rows);
    if (c->n4_rows < upto_rows) {
        const long long n = upto_rows - c->n4_rows;
        const long long blocks = std::min<long long>((n * 16 + 255) / 256, 256 * 32);
        auto kern = c->vtype == VG_TYPE_U8 ? vg_to_n4_kernel<T_U8> : vg_to_n4_kernel<T_I8>;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_rows, (long long)c->n4_rows, n,
                           (long long)c->stride, c->dim, c->d_rows_n4, ns, reinterpret_cast<VgN4Stat *>(c->d_n4stat));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return vg_fail(VG_ERR_HIP, "nibble shadow pass failed: %s", hipGetErrorString(e));
        c->n4_rows = upto_rows;
    }
    return VG_OK;
}

static long long q8_shadow_stride(const vg_corpus *c) { return (((long long)c->dim + 15) / 16) * 16; }
// What the filter scans stream.  Default: the int8 shadow copy - a quarter of an f32 corpus' bytes (+ 26 % HBM), half of an
// f16 / bf16 corpus' (+ 52 %).  VG_SCAN_FILTER_SHADOW=bf16 (or "rows"): f32 corpora through the bf16 shadow copy (half the
// bytes, + 50 % HBM - the copy the batched filter kernel reads), f16 / bf16 corpora through their own rows (no copy).
// L1 (f16 / bf16 only) has no dot-product bound: always the rows.
static bool filter_uses_q8(const vg_corpus *c, int metric) {
    if (c->vtype != VG_TYPE_F32 && c->vtype != VG_TYPE_F16 && c->vtype != VG_TYPE_BF16) return false;
    if (metric == VG_DIST_L1 || c->q8_disabled) return false;
    const int e = vg_sw(SW_VG_SCAN_FILTER_SHADOW, 0);                 // (its first letter)
    return !(e == 'b' || e == 'B' || e == 'r' || e == 'R');
}
int vg_ensure_q8_shadow(vg_corpus *c) {
    const long long qs = q8_shadow_stride(c);
    if (c->q8_cap < c->n_rows) {
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rows_q8) hipFree(c->d_rows_q8);
        if (c->d_q8stat) hipFree(c->d_q8stat);
        c->d_rows_q8 = nullptr; c->d_q8stat = nullptr; c->q8_cap = 0; c->q8_rows = 0;
        if (hipMalloc(&c->d_rows_q8, (size_t)cap * qs) != hipSuccess || hipMalloc(&c->d_q8stat, (size_t)cap * sizeof(float2)) != hipSuccess) {
            (void)hipGetLastError();
            if (c->d_rows_q8) hipFree(c->d_rows_q8);
            if (c->d_q8stat) hipFree(c->d_q8stat);
            c->d_rows_q8 = nullptr; c->d_q8stat = nullptr;
            return vg_fail(VG_ERR_NOMEM, "no device memory for the int8 shadow copy (%lld rows)", (long long)cap);
        }
        c->q8_cap = cap;
    }
    if (c->q8_rows < c->n_rows) {
        const long long n = c->n_rows - c->q8_rows;
        const long long blocks = std::min<long long>((n * 16 + 255) / 256, 256 * 32);
        // rows of up to 128 chunks: the single-read kernel (the row lives in registers between its two sweeps)
        to_q8_fn_t reg = vg_sw(SW_VG_Q8_TWO_READS, 0) ? nullptr
                       : (c->vtype == VG_TYPE_F32 ? pick_to_q8_reg<T_F32>(c->nch) : (c->vtype == VG_TYPE_F16 ? pick_to_q8_reg<T_F16>(c->nch) : pick_to_q8_reg<T_BF16>(c->nch)));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c->profiling) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, c->stream); }
        if (reg) {
            hipLaunchKernelGGL(reg, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_rows, (long long)c->q8_rows, n,
                               (long long)c->stride, c->dim, c->nch, c->d_rows_q8, qs, reinterpret_cast<float2 *>(c->d_q8stat));
        } else {
            auto kern = c->vtype == VG_TYPE_F32 ? vg_to_q8_kernel<T_F32> : (c->vtype == VG_TYPE_F16 ? vg_to_q8_kernel<T_F16> : vg_to_q8_kernel<T_BF16>);
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_rows, (long long)c->q8_rows, n,
                               (long long)c->stride, c->dim, c->d_rows_q8, qs, reinterpret_cast<float2 *>(c->d_q8stat));
        }
        if (e0) {                                            // (profiling runs only: what bench.py --workload stage reports)
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&c->pass_ms[2], e0, e1);
            c->pass_rows[2] = n;
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return vg_fail(VG_ERR_HIP, "int8 shadow pass failed: %s", hipGetErrorString(e));
        c->q8_rows = c->n_rows;
    }
    return VG_OK;
}
static int filter_mode_of(int metric) {
    return (metric == VG_DIST_DOT) ? VGF_DOT : (metric == VG_DIST_COSINE ? VGF_COS : (metric == VG_DIST_L1 ? VGF_L1 : VGF_L2));
}
// chunks per lane the filter kernels hold without spilling under the 128-VGPR cap of 16 wavefronts per CU (tools/kernel_regs.py)
static int filter_u_cap(const vg_corpus *) { return 6; }
// bytes per row the filter streams: the bf16 shadow copy of an f32 corpus, the rows themselves otherwise
static long long filter_stream_stride(const vg_corpus *c, int metric) {
    if (c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8) return n4_shadow_stride(c);
    if (filter_uses_q8(c, metric)) return q8_shadow_stride(c);
    return c->vtype == VG_TYPE_F32 ? vg_bf16_shadow_stride(c) : (long long)c->stride;
}

// Returns -1 when the shape is not served (caller takes the plain scan).
int vg_launch_scan_filter(vg_corpus *c, int metric, const uint8_t *dev_query, int k, uint64_t *dev_out_keys, hipStream_t stream,
                          bool ref_emit, uint64_t *final_out) {
    if (!scan_filter_serves(c, metric)) return -1;
    const bool f32 = (c->vtype == VG_TYPE_F32);
    // The shadow copy and the norms cost HBM next to the corpus (see filter_uses_q8).  An f32 corpus they do not fit next to
    // keeps the plain f32 scan (which served it before the filter existed) instead of failing every query; an f16 / bf16
    // corpus falls back to filtering over its own rows.
    const bool n4 = (c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8);
    // uint8 / int8, not switched on explicitly: this scan PROBES (filter over a prefix, count the candidates, decide, then answer
    // through whichever kernel the decision names - see scan_filter_serves_n4)
    const bool probing = n4 && !n4_explicit(c) && c->n4_probe != 1;
    const int64_t scan_rows = probing ? std::min<int64_t>(c->n_rows, 1ll << 21) : c->n_rows;
    int rc = n4 ? VG_OK : vg_ensure_row_norms(c);
    bool q8 = filter_uses_q8(c, metric);
    if (n4) {
        rc = vg_ensure_n4_shadow(c, scan_rows);
        if (rc == VG_ERR_NOMEM) { (void)hipGetLastError(); c->n4_disabled = true; return -1; }
    } else if (rc == VG_OK && q8) {
        rc = vg_ensure_q8_shadow(c);
        if (rc == VG_ERR_NOMEM && !f32) { (void)hipGetLastError(); c->q8_disabled = true; q8 = false; rc = VG_OK; }
    } else if (rc == VG_OK && f32) rc = vg_ensure_bf16_shadow(c);
    if (rc == VG_ERR_NOMEM) {
        (void)hipGetLastError();                         // clear the sticky allocation error
        c->filter_disabled = true;
        return -1;
    }
    if (rc != VG_OK) return rc;
    const long long bs = filter_stream_stride(c, metric);
    const int nch_b = (int)(bs / 16);
    Shape s;
    vg_choose_shape(nch_b, VG_TYPE_U8, A_DOT, &s, filter_u_cap(c));
    if (s.long_rows) return -1;
    {   // experiment override of the filter's own launch shape
        const int fl = vg_sw(SW_VG_FILTER_LPR_LOG2, -1), fu = vg_sw(SW_VG_FILTER_U, -1);
        if (fl >= 0 && fl <= 6 && fu > 0 && (nch_b + (1 << fl) - 1) / (1 << fl) <= fu &&
            (n4 ? pick_n4<true>(c->vtype, filter_mode_of(metric), fu) : pick_filter<true>(c->vtype, filter_mode_of(metric), fu, q8))) { s.lpr_log2 = fl; s.U = fu; }
    }
    Shape xs;                                            // the plain kernel's own shape: the exact evaluation sums in its order
    vg_plain_scan_shape(c, metric, &xs);
    if (xs.long_rows) return -1;
    if ((rc = vg_ensure_filter_counters(c)) != VG_OK) return rc;
    {   // Selectivity guard.  The bound cannot separate rows that are (nearly) identical to each other: on such data every row
        // is a candidate and the exact evaluations - serial per wavefront - cost more than the plain scan.  The kernels count
        // them, a copy behind every launch mirrors the counter into pinned host memory; when the completed launches since the
        // last look averaged more than 1/8 of the rows (an exact evaluation is cheap since up to 64 / xlpr of them run at once -
        // ~0.2 ns each chip-wide - and the filter pass itself takes a quarter to a half of the plain scan's time), the next 256 scans of this corpus take the plain kernel, then the
        // filter is tried again.
        // (the mirror holds the counter as of the last launch whose copy has LANDED and, in word [2], how many filter launches had
        // finished by then: averaging over launches still in flight would bias the figure low)
        // (the two words are written by different threads of the merge's workgroup: read [2], [0], [2] again and retry on a mismatch,
        // so that a copy landing in between cannot pair one launch's count with the next launch's number)
        unsigned long long now = 0;
        long long landed = 0;
        for (int attempt = 0; attempt < 4; ++attempt) {
            landed = (long long)*(volatile unsigned long long *)(c->h_filter_evals + 2);
            now = *(volatile unsigned long long *)c->h_filter_evals;
            if (landed == (long long)*(volatile unsigned long long *)(c->h_filter_evals + 2)) break;
        }
        const long long launches = std::min<long long>(c->filter_launches, landed) - c->filter_launches_seen;
        if (launches >= 2) {
            if ((now - c->filter_evals_seen) / (unsigned long long)launches > (unsigned long long)(c->n_rows / 8) && !vg_sw(SW_VG_SCAN_FILTER_NO_GUARD, 0))
                c->filter_cooldown = 256;
            // the pre-pass' share of the rows follows the same average: many candidates (a loose bound on this data) are worth a
            // tighter start threshold - a longer pre-pass; few are not (measured, profiles/r3m: nibble filter on the C3 bytes,
            // 295k candidates per query at 1/128, 0.90 ms -> 83k at 1/16, 0.80 ms; int8 filter on C2, 6k candidates: 1/128 best)
            const unsigned long long avg = (now - c->filter_evals_seen) / (unsigned long long)launches;
            if (avg > (unsigned long long)(c->n_rows / 128)) c->filter_prepass_div = std::max(16, c->filter_prepass_div / 2);
            else if (avg < (unsigned long long)(c->n_rows / 1024)) c->filter_prepass_div = std::min(128, c->filter_prepass_div * 2);
            c->filter_evals_seen = now;
            c->filter_launches_seen += launches;
        }
        if (c->filter_cooldown > 0 && !probing) { --c->filter_cooldown; return -1; }
    }
    unsigned long long evals_before = 0;
    if (probing) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(&evals_before, c->d_filter_evals, sizeof(evals_before), hipMemcpyDeviceToHost));
    }
    if (stream != c->stream) {                           // both passes (and the memset) ran on the corpus stream
        if (!c->norm_ev) HIP_TRY(hipEventCreateWithFlags(&c->norm_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->norm_ev, c->stream));
        HIP_TRY(hipStreamWaitEvent(stream, c->norm_ev, 0));
    }
    const bool nt = (vg_sw(SW_VG_NT, -1) >= 0) ? vg_sw(SW_VG_NT, -1) != 0 : (scan_rows * bs > (256ll << 20));
    const int mode = filter_mode_of(metric);
    filter_fn_t fn = n4 ? (nt ? pick_n4<true>(c->vtype, mode, s.U) : pick_n4<false>(c->vtype, mode, s.U))
                        : (nt ? pick_filter<true>(c->vtype, mode, s.U, q8) : pick_filter<false>(c->vtype, mode, s.U, q8));
    if (!fn) return -1;
    const int rpb = VG_WAVE >> s.lpr_log2;
    const long long nbatch = (scan_rows + rpb - 1) / rpb;
    long long blocks = (nbatch + VG_WAVES_PER_BLOCK - 1) / VG_WAVES_PER_BLOCK;
    blocks = std::max<long long>(1, std::min<long long>(blocks, (long long)c->cu_count));
    blocks = std::min<long long>(blocks, VG_SEL_MAX_HEADS);
    FilterScanArgs a{};
    a.shadow = n4 ? c->d_rows_n4 : (q8 ? c->d_rows_q8 : (f32 ? c->d_rows_bf : c->d_rows));
    a.q8stat = reinterpret_cast<const float2 *>(n4 ? c->d_n4stat : c->d_q8stat); a.rows = c->d_rows; a.query = dev_query; a.row_norm = c->d_xnorm; a.cand = c->d_cand;
    a.n_rows = scan_rows; a.stride = c->stride; a.bstride = bs; a.nch = c->nch; a.nch_b = nch_b;
    a.lpr_log2 = s.lpr_log2; a.k = k; a.root = (metric == VG_DIST_L2) ? 1 : 0; a.dim = c->dim;
    a.mode = mode;
    // |s~ - s| <= cerr |q||x| (vg_scan_filter.h): (D + 64) 2^-21 for the f32 sums; an f32 corpus is read through its bf16
    // shadow copy - BOTH factors of every product rounded, unit roundoff u = 2^-8 each: + (1+u)^2 - 1 = 2^-7 + 2^-16
    a.cerr = (float)(c->dim + 64) * 4.76837158203125e-7f + (f32 ? 0.0078125f + 1.52587890625e-5f : 0.0f);
#ifdef VG_TEST_ROUND1_CERR     // tools/build_round1_cerr_variant.sh only: round 1's unsound constant, to show tests/test_gpu_filter_bound.py red on it
    if (f32) a.cerr = 0.00390625f + 1.6e-5f + (float)(c->dim + 64) * 4.76837158203125e-7f;
#endif
    a.rel = (float)(c->dim + 64) * 2.384185791015625e-7f;
    a.xlpr_log2 = xs.lpr_log2; a.xU = xs.U;
    a.evals = c->d_filter_evals;
    // the staged query; behind it, for an int8 shadow of f16 / bf16 rows, its exact f32 copy (16 * nch_b floats)
    const size_t smem = std::max<size_t>((size_t)c->nch * 16 + ((q8 && !f32) ? (size_t)nch_b * 64 : 0), (size_t)VG_PUBLISH_LDS_BYTES);
    if (c->append_pending && stream != c->stream) HIP_TRY(hipStreamWaitEvent(stream, c->append_ev, 0));
    // Pre-pass: a plain scan of the first 1/64 of the rows.  Its k-th best distance bounds the final k-th best from
    // above, so no wavefront has to warm its list up from +Inf (k ln(rows per wavefront / k) exact evaluations each,
    // ~0.35 ms in all); the filter scan below still covers every row.
    // tie_order = reference (ref_emit): the pre-pass doubles as the replay's prefix pass (it also stores its rows' distances) and the
    // filter kernel emits the later rows that can enter the reference's slots - a probing launch never does (its answer is discarded)
    const bool emitting = ref_emit && !probing && scan_rows >= VG_REF_EMIT_MIN_ROWS;
    const bool prepass = (vg_sw(SW_VG_SCAN_FILTER_PREPASS, 1) != 0 && scan_rows >= (1 << 20)) || emitting;
    hipEvent_t *evs = probing ? nullptr : vg_prof_slot(c, (uint8_t)(VG_EVF_MERGE | (prepass ? VG_EVF_PREPASS : 0)));
    if (evs) hipEventRecord(evs[0], stream);
    a.init_keys = nullptr;
    a.init_lists = nullptr;
    a.n_init_lists = 0;
    a.emit = nullptr;
    a.emit_cap = 0;
    if (prepass) {
        ScanPlan pre;
        // (1/128 of the rows: 38 us instead of 60 at 10M x 384 for ~2x the exact evaluations of the 0.6 ms pass - measured, profiles/r2y)
        pre.n_rows = std::max<int64_t>(65536, scan_rows / std::max(1, vg_sw(SW_VG_SCAN_FILTER_PREPASS_DIV, c->filter_prepass_div)));
        if (scan_rows < (1 << 20)) pre.n_rows = vg_ref_prefix_for(scan_rows);      // (a pre-pass only because of the replay)
        pre.allow_filter = false;
        pre.record = false;
        if (emitting) {
            pre.n_rows = std::min<int64_t>(pre.n_rows, VG_REF_PREFIX_MAX);
            const int rcb = vg_ensure_ref_buffers(c, pre.n_rows);
            if (rcb != VG_OK) return rcb;
            pre.store_prefix = c->d_ref_prefix;
            pre.emit_reset = c->d_below;
            a.emit = c->d_below;
            a.emit_cap = VG_BELOW_CAP;
            c->ref_prefix_rows = pre.n_rows;
        }
        // The pre-pass' per-CU lists go to the filter kernel UNMERGED (every workgroup takes the k-th smallest list head itself,
        // vg_kth_head): one launch less on the query's critical path (pre-pass 37 -> 26 us, the filter kernel + 4 us: 0.684 -> 0.676 ms per query).  VG_SCAN_FILTER_PREMERGE=1: round 2's
        // form (merge launch, init_keys) - also what serves a staged query too long to leave the head scratch free.
        int n_pre_lists = 0;
        const bool unmerged = !vg_sw(SW_VG_SCAN_FILTER_PREMERGE, 0) &&
                              smem >= (size_t)VG_PUBLISH_LDS_BYTES &&
                              (size_t)c->nch * 16 + ((q8 && !f32) ? (size_t)nch_b * 64 : 0) + 64 <= (size_t)VG_PUBLISH_LDS_BYTES - VG_KTH_HEAD_SCRATCH_BYTES - 16;
        if (unmerged) {
            if (!c->d_cand_pre) HIP_TRY(hipMalloc(&c->d_cand_pre, (size_t)VG_SEL_MAX_HEADS * VG_WAVE * sizeof(uint64_t)));
            pre.lists_out = c->d_cand_pre;
            pre.n_lists_out = &n_pre_lists;
        }
        const int rcp = vg_launch_plain_scan(c, metric, dev_query, k, dev_out_keys, stream, pre);
        if (rcp != VG_OK) return rcp;
        if (unmerged) { a.init_lists = c->d_cand_pre; a.n_init_lists = n_pre_lists; }
        else a.init_keys = dev_out_keys;                 // read by every workgroup before the final merge overwrites it
        if (evs) hipEventRecord(evs[1], stream);
    }
    if (smem > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(VG_BLOCK), smem, stream, a);
    ++c->filter_launches;
    if (evs) hipEventRecord(evs[2], stream);
    // counter [0] and the number of finished filter launches [2] (bumped by the kernel itself) travel together: consistent.  The
    // merge's workgroup copies them into the pinned mirror on its way (VG_SCAN_FILTER_MIRROR_COPY=1: a copy command behind the
    // merge, the earlier form; sending that down a side stream behind the filter kernel was measured too: the event record + wait
    // cost more than the copy holds up the key read-back - 0.684 against 0.676 ms per query, profiles/r4v_filter_floor_ab.txt).
    const bool mirror_in_merge = !probing && vg_sw(SW_VG_SCAN_FILTER_MIRROR_COPY, 0) == 0;
    const int rcm = vg_launch_merge_one((const uint64_t *)c->d_cand, (int)blocks, k, (final_out && !probing) ? final_out : dev_out_keys, stream,
                                        mirror_in_merge ? c->d_filter_evals : nullptr, mirror_in_merge ? c->h_filter_evals : nullptr);
    if (evs) hipEventRecord(evs[3], stream);
    if (!mirror_in_merge) HIP_TRY(hipMemcpyAsync(c->h_filter_evals, c->d_filter_evals, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    if (rcm != 0) return vg_fail(VG_ERR_HIP, "merge launch failed: %s", hipGetErrorString((hipError_t)rcm));
    HIP_TRY(hipGetLastError());
    if (probing) {
        // the probe's own answer (the prefix' top k) is discarded: count what it evaluated exactly and decide.  1 candidate per
        // 8 prefix rows - the guard's own limit - is inside the filter's break-even (~1 in 5 at 0.2 ns per evaluation).
        HIP_TRY(hipStreamSynchronize(stream));
        unsigned long long evals_after = 0;
        HIP_TRY(hipMemcpy(&evals_after, c->d_filter_evals, sizeof(evals_after), hipMemcpyDeviceToHost));
        const bool selective = (evals_after - evals_before) * 8ull < (unsigned long long)scan_rows;
        c->n4_probe = selective ? 1 : 2;
        c->n4_probe_rows = c->n_rows;
        c->filter_evals_seen = evals_after;              // (the guard's averages start after the probe)
        c->filter_launches_seen = c->filter_launches;
        if (!selective) {                                // nothing is kept of an unselective corpus' prefix copy
            hipFree(c->d_rows_n4); hipFree(c->d_n4stat);
            c->d_rows_n4 = nullptr; c->d_n4stat = nullptr; c->n4_rows = 0; c->n4_cap = 0;
        }
        return vg_launch_scan_filter(c, metric, dev_query, k, dev_out_keys, stream, ref_emit, final_out);   // 1: the filter over every row; 2: -1 (plain scan)
    }
    return VG_OK;
}

bool vg_scan_filter_would_serve(const vg_corpus *c, int metric, int k) {
    if (k > VG_MAX_FUSED_K || c->filter_cooldown > 0 || !scan_filter_serves(c, metric)) return false;
    if ((c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8) && !n4_explicit(c) && c->n4_probe != 1) return false;   // (not probed yet / not selective)
    return true;
}

static const char *filter_type_tag(int t) { return t == VG_TYPE_F32 ? "f32" : (t == VG_TYPE_F16 ? "f16" : (t == VG_TYPE_BF16 ? "bf16" : (t == VG_TYPE_U8 ? "u8" : "i8"))); }

// "scan_filter_<type>_<metric>[_bf16]_u<U>_lpr<L>[_nt]" when the filter serves (corpus, metric); false otherwise
bool vg_scan_filter_name(vg_corpus *c, int metric, char *out, size_t out_len) {
    if (!scan_filter_serves(c, metric)) return false;
    const long long bs = filter_stream_stride(c, metric);
    Shape fs, xs;
    vg_choose_shape((int)(bs / 16), VG_TYPE_U8, A_DOT, &fs, filter_u_cap(c));
    vg_plain_scan_shape(c, metric, &xs);
    if (fs.long_rows || xs.long_rows) return false;
    const bool nt = (vg_sw(SW_VG_NT, -1) >= 0) ? vg_sw(SW_VG_NT, -1) != 0 : (c->n_rows * bs > (256ll << 20));
    static const char *mtag[4] = {"l2", "dot", "cos", "l1"};
    snprintf(out, out_len, "scan_filter_%s_%s%s_u%d_lpr%d%s", filter_type_tag(c->vtype), mtag[filter_mode_of(metric)],
             (c->vtype == VG_TYPE_U8 || c->vtype == VG_TYPE_I8) ? "_n4" : (filter_uses_q8(c, metric) ? "_q8" : (c->vtype == VG_TYPE_F32 ? "_bf16" : "")),
             fs.U, 1 << fs.lpr_log2, nt ? "_nt" : "");
    return true;
}
