// vg_multi.hip - host side of the multi-query scan (vg_scan_multi.h): shape choice, kernel table, launch.
// A translation unit of its own so that its ~100 kernel instances compile next to vg_api.hip's, not after them.
#include "vg_internal.h"

#include "vg_scan_multi.h"

typedef void (*scan_fn_t)(ScanArgs);

template <int VT, int ACC, int NQ>
static scan_fn_t pick_multi_u(int U) {
    if constexpr (NQ == 4) {
        switch (U) {
            case 1: return vg_scan_multi_kernel<VT, ACC, 1, 4, true>;
            case 2: return vg_scan_multi_kernel<VT, ACC, 2, 4, true>;
            case 3: return vg_scan_multi_kernel<VT, ACC, 3, 4, true>;
        }
    } else {
        switch (U) {
            case 4: return vg_scan_multi_kernel<VT, ACC, 4, 2, true>;
            case 6: return vg_scan_multi_kernel<VT, ACC, 6, 2, true>;
        }
    }
    return nullptr;
}
template <int VT, int NQ>
static scan_fn_t pick_multi_acc(int acc, int U) {
    switch (acc) {
        case A_L2: return pick_multi_u<VT, A_L2, NQ>(U);
        case A_COS: return pick_multi_u<VT, A_COS, NQ>(U);
        case A_DOT: return pick_multi_u<VT, A_DOT, NQ>(U);
        case A_L1: return pick_multi_u<VT, A_L1, NQ>(U);
    }
    return nullptr;
}
static scan_fn_t pick_multi(int vtype, int acc, int U, int NQ) {
    switch (vtype) {
        case VG_TYPE_F32: return NQ == 4 ? pick_multi_acc<T_F32, 4>(acc, U) : pick_multi_acc<T_F32, 2>(acc, U);
        case VG_TYPE_U8: return NQ == 4 ? pick_multi_acc<T_U8, 4>(acc, U) : pick_multi_acc<T_U8, 2>(acc, U);
        case VG_TYPE_I8: return NQ == 4 ? pick_multi_acc<T_I8, 4>(acc, U) : pick_multi_acc<T_I8, 2>(acc, U);
    }
    return nullptr;
}

// (queries per pass, launch shape) of the multi-query scan for this corpus / metric; 0 when there is none
static int multi_plan(const vg_corpus *c, int metric, VgShape *s) {
    const int acc = vg_metric_to_acc(metric);
    if (acc < 0) return 0;
    // f16 / bf16 scans are bound by their f64 accumulation (the reference's arithmetic), not by HBM: two queries per
    // pass measured 0.75x - 1.2x of two single scans (and spill at U = 3), so they keep the single-query kernel
    if (c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16) return 0;
    vg_choose_shape(c->nch, c->vtype, acc, s, 3);
    if (!s->long_rows && s->U <= 3) return 4;
    vg_choose_shape(c->nch, c->vtype, acc, s, 6);
    if (!s->long_rows && (s->U == 4 || s->U == 6)) return 2;
    return 0;
}

int vg_multi_queries_per_pass(const vg_corpus *c, int metric) {
    VgShape s;
    return multi_plan(c, metric, &s);
}

// NQ = vg_multi_queries_per_pass() queries (zero-padded rows of the corpus stride, back to back at dev_queries) against
// the corpus in ONE pass; dev_cand: NQ * (<= 256) * 64 keys of scratch; dev_out_keys: NQ x 64 keys.  Asynchronous on
// `stream`.  Returns -1 when the shape has no multi-query kernel, VG_OK or an error code otherwise.
int vg_launch_scan_multi(vg_corpus *c, int metric, const uint8_t *dev_queries, int k, uint64_t *dev_cand, uint64_t *dev_out_keys,
                         hipStream_t stream) {
    if (k < 1 || k > VG_MAX_FUSED_K) return -1;
    VgShape s;
    const int NQ = multi_plan(c, metric, &s);
    if (NQ == 0) return -1;
    const int acc = vg_metric_to_acc(metric);
    scan_fn_t fn = pick_multi(c->vtype, acc, s.U, NQ);
    if (!fn) return -1;
    const int rpb = VG_WAVE >> s.lpr_log2;
    const long long nbatch = (c->n_rows + rpb - 1) / rpb;
    long long blocks = (nbatch + VG_WAVES_PER_BLOCK - 1) / VG_WAVES_PER_BLOCK;
    blocks = std::max<long long>(1, std::min<long long>(blocks, (long long)c->cu_count));
    blocks = std::min<long long>(blocks, VG_SEL_MAX_HEADS);
    if (c->append_pending && stream != c->stream) HIP_TRY(hipStreamWaitEvent(stream, c->append_ev, 0));
    ScanArgs a{};
    a.rows = c->d_rows; a.query = dev_queries; a.cand = dev_cand; a.out_dist = nullptr; a.n_rows = c->n_rows;
    a.stride = c->stride; a.nch = c->nch; a.lpr_log2 = s.lpr_log2; a.k = k; a.root = (metric == VG_DIST_L2) ? 1 : 0;
    a.dim = c->dim; a.row_nn = nullptr; a.store_lds_off = 0;
    const size_t smem = std::max<size_t>((size_t)NQ * c->nch * 16, (size_t)VG_PUBLISH_LDS_BYTES);
    if (smem > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(VG_BLOCK), smem, stream, a);
    int rc = vg_launch_merge(dev_cand, (int)blocks, k, dev_out_keys, NQ, stream);
    if (rc != 0) return vg_fail(VG_ERR_HIP, "merge launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIP_TRY(hipGetLastError());
    return VG_OK;
}
