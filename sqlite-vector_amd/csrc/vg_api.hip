// vg_api.hip - the scan entry points of the C-ABI (include/vectorgpu.h) on top of the gfx950 kernels of vg_scan.h.
//
// Host-side responsibilities only: device memory for the staged corpus, query upload, kernel selection and
// launch, decoding the k winning keys into (rowid, distance).  No distance is ever computed on the host: if
// the HIP runtime / a gfx950 device is missing every entry point fails with VG_ERR_NO_DEVICE.
#include "vg_internal.h"

#include "vg_scan.h"

// ------------------------------------------------------------------------------------------------ kernel selection

typedef void (*scan_fn_t)(ScanArgs);
scan_fn_t vg_pick_scan_kernel_ex(int vtype, int acc, int U);      // vg_scan_ex.hip

typedef VgShape Shape;

static const int kAllowedU[] = {1, 2, 3, 4, 6, 8};

// Pick (lanes per row, chunks per lane): cover nch chunks with lpr*U slots, wasting as few lane slots as
// possible; prefer >= 128 contiguous bytes per row per load instruction, then U = 6/4/3 (bytes in flight per lane).
// Rows that no (lpr <= 64, U <= cap) shape covers take the long-row kernel (query in LDS, one row per wavefront).
// f16 / bf16: the f64 arithmetic (4 f64 accumulators + the widening temporaries) leaves room for fewer chunks per lane
// under the 128-VGPR cap of 16 wavefronts per CU.  The query stays in registers as RAW halves (vg_scan.h launders it
// every batch so the compiler cannot hoist widened copies out of the loop); with that U = 6 fits for everything but
// bf16 dot / cosine, which spill beyond U = 3 (measured 2.4 TB/s at U = 6).
static int max_chunks_per_lane(int vtype, int acc, bool bf16_l2_u3) {
    if (vtype == VG_TYPE_F16) return 6;
    // bf16: dot / cosine spill beyond 3.  L2 / L1 fit 6; with the unconditional batch loads L2 measures 2 % faster at 8 lanes x 6
    // (10M x 384: 6.55 / 6.69 against 6.43 / 6.57 TB/s in two alternating passes), L1 2 % slower (profiles/r5d_shape_ab_f32_half.txt)
    if (vtype == VG_TYPE_BF16) return (acc == A_L2 && !bf16_l2_u3) ? 6 : 3;
    return 8;
}

// preference among shapes that waste the same share of lane slots (higher is better).  Round 1 preferred 6 chunks per lane; round 3
// re-measured (profiles/r4k_launch_shape_sweep_c2_c3.txt, r4j_c4_one_gpu_launch_shapes.txt): for f32 / uint8 / int8 rows TWICE the
// lanes per row with 3 chunks each - 512 contiguous bytes of a row per load instruction instead of 256 - streams faster at every
// size: 10M x 384 f32 0.863 -> 0.875 of the peak, 12.5M 0.854 -> 0.870, 100M 0.827 -> 0.858, 10M x 768 uint8 0.809 -> 0.813.
// 64 lanes x 3 beats 32 x 6 as well (2.6M x 768 f32: 6.72 -> 6.92 TB/s, the two crossbar steps of its butterfly notwithstanding),
// while 4 chunks per lane are NOT improved by 2 at twice the lanes (f32 128 / 256 / 512, uint8 512 / 1024: equal or slower).
// f16 too since the batch loads are unconditional (vg_load_batch): while a branch around every load cost the prefetch its overlap,
// 8 x 6 - more bytes per wait - hid that best and 16 x 3 measured slower (profiles/r4b_kernel_matrix_half_shapes.txt); with the
// prefetch really in flight behind the arithmetic 16 x 3 wins on every metric (10M x 384 f16: L2 6.91 -> 6.91, cosine 6.53 -> 6.85,
// dot 6.54 -> 7.01, L1 6.90 -> 6.97 TB/s; 5M x 768: 6.34-6.54 -> 6.65-6.77, profiles/r5c_shape_sweep_unconditional_loads.txt).
// bf16 takes at most 3 chunks per lane anyway (max_chunks_per_lane).
// the experiment switches of the shape choice, read ONCE per choice (they sat inside its loops: ~20 getenv calls per scan launch on a
// path that is tuned to ~34 us per query)
struct ShapeEnv { bool f16_round3, pref_round1, bf16_l2_u3, int_short_round3, force_long; int lpr_log2, u; };
static ShapeEnv shape_env() {
    ShapeEnv e;
    e.f16_round3 = vg_sw(SW_VG_SHAPE_F16_ROUND3, 0) != 0;
    e.pref_round1 = vg_sw(SW_VG_SHAPE_PREF_ROUND1, 0) != 0;
    e.bf16_l2_u3 = vg_sw(SW_VG_SHAPE_BF16_L2_U3, 0) != 0;
    e.int_short_round3 = vg_sw(SW_VG_SHAPE_INT_SHORT_ROUND3, 0) != 0;
    e.force_long = vg_sw(SW_VG_FORCE_LONG, 0) != 0;
    e.lpr_log2 = vg_sw(SW_VG_LPR_LOG2, -1);
    e.u = vg_sw(SW_VG_U, -1);
    return e;
}
static int shape_pref(const ShapeEnv &env, int vtype, int l2, int U, bool ragged) {
    static const int pref[9] = {0, 1, 2, 4, 5, 0, 6, 0, 3};
    const bool wide = (vtype == VG_TYPE_F32 || vtype == VG_TYPE_U8 || vtype == VG_TYPE_I8 || (vtype == VG_TYPE_F16 && !env.f16_round3)) &&
                      !env.pref_round1;
    if (wide && U == 3) return 7;
    // rows no shape covers exactly (e.g. 100 floats = 25 chunks): 2 chunks per lane at twice the lanes beat 4 (longer contiguous
    // runs over rows that are not line-aligned): 15M x 100 f32 5.3 -> 6.1-6.6 TB/s (profiles/r4q_short_rows_shape_ab.txt)
    if (wide && ragged && U == 2 && l2 <= 4) return 6;      // (not across 32+ lanes: 7.5M x 200 f32 cosine 6.6 -> 6.0 TB/s with the crossbar step)
    return pref[U];
}

bool vg_choose_shape(int nch, int vtype, int acc, VgShape *out, int u_cap) {
    const ShapeEnv env = shape_env();
    const int max_u = std::min(max_chunks_per_lane(vtype, acc, env.bf16_l2_u3), u_cap);
    const bool round1 = env.pref_round1;
    // Short rows (3 .. 8 chunks): with ONE chunk per lane a batch is one load + a whole epilogue (butterfly, conversions, sqrt /
    // divide, key, ballot) for 64 / lanes-per-row rows - half the lanes per row with 2 chunks each amortise it over twice the rows.
    // f32 (32 floats: 5.9 -> 6.2 TB/s on every metric) and - re-measured with the unconditional batch loads,
    // profiles/r5k_shape_sweep_short_rows.txt - f16 / bf16, whose f64 epilogue is the heaviest (64 halves: L2 4.8 -> 5.9, cosine 4.4 -> 5.8,
    // dot 5.7 -> 6.1 TB/s).  NOT uint8 / int8 any more: the rule had been adopted for their L2 / cosine while the prefetch did not
    // overlap; with it in flight one chunk per lane wins on every metric (64 bytes L2 5.4 -> 6.0, 100 bytes 4.3 -> 5.0, 128 bytes 5.4 -> 6.1).
    const bool short_rows = !round1 && nch >= 3 && nch <= 8 &&
                            (vtype == VG_TYPE_F32 || vtype == VG_TYPE_F16 || vtype == VG_TYPE_BF16 ||
                             (env.int_short_round3 && (vtype == VG_TYPE_U8 || vtype == VG_TYPE_I8) && (acc == A_L2 || acc == A_COS)));
    double best_eff = -1.0;
    for (int l2 = 0; l2 <= 6; ++l2) {                        // (the best cover any shape reaches: "ragged" rows have none at 1.0)
        const int lpr = 1 << l2, need = (nch + lpr - 1) / lpr;
        for (int a : kAllowedU) if (a >= need && a <= max_u) { best_eff = std::max(best_eff, (double)nch / ((double)lpr * a)); break; }
    }
    const bool ragged = best_eff < 0.999;
    best_eff = -1.0; int best_flag = -1, best_pref = -1; Shape best = {0, 0, false};
    for (int l2 = 0; l2 <= 6; ++l2) {
        int lpr = 1 << l2;
        int need = (nch + lpr - 1) / lpr;
        int U = 0;
        for (int a : kAllowedU) if (a >= need && a <= max_u) { U = a; break; }
        if (!U) continue;
        double eff = (double)nch / ((double)lpr * U);
        int flag = (lpr >= 8 || lpr >= nch || (short_rows && U == 2)) ? 1 : 0;
        const int pr = shape_pref(env, vtype, l2, U, ragged);
        bool better = eff > best_eff + 1e-9 ||
                      (fabs(eff - best_eff) <= 1e-9 && (flag > best_flag || (flag == best_flag && pr > best_pref)));
        if (better) { best_eff = eff; best_flag = flag; best_pref = pr; best.lpr_log2 = l2; best.U = U; }
    }
    if (best.U == 0 || env.force_long) { best.lpr_log2 = 6; best.U = VG_LONG_U; best.long_rows = true; }
    int fl = env.lpr_log2, fu = env.u;   // experiment overrides
    if (!best.long_rows && fl >= 0 && fu > 0 && (nch + (1 << fl) - 1) / (1 << fl) <= fu) { best.lpr_log2 = fl; best.U = fu; }
    *out = best;
    return true;
}
static bool choose_shape(int nch, int vtype, int acc, Shape *out) { return vg_choose_shape(nch, vtype, acc, out, 8); }
int vg_metric_to_acc(int metric);
extern "C" int vg_plan_scan_shape(int vtype, int dim, int metric, int *lanes_per_row, int *chunks_per_lane, int *long_rows) {
    const int acc = vg_metric_to_acc(metric);
    if (vtype < VG_TYPE_F32 || vtype > VG_TYPE_I8 || dim < 1 || acc < 0) return vg_fail(VG_ERR_INVALID, "vg_plan_scan_shape: bad type / dim / metric");
    Shape s;
    choose_shape((int)(((long long)dim * vg_elem_size(vtype) + 15) / 16), vtype, acc, &s);
    if (lanes_per_row) *lanes_per_row = 1 << s.lpr_log2;
    if (chunks_per_lane) *chunks_per_lane = s.U;
    if (long_rows) *long_rows = s.long_rows ? 1 : 0;
    return VG_OK;
}

template <int VT, int ACC, bool NT>
static scan_fn_t pick_u(int U) {
    switch (U) {
        case 1: return vg_scan_kernel<VT, ACC, 1, NT>;
        case 2: return vg_scan_kernel<VT, ACC, 2, NT>;
        case 3: return vg_scan_kernel<VT, ACC, 3, NT>;
        case 4: return vg_scan_kernel<VT, ACC, 4, NT>;
        case 6: return vg_scan_kernel<VT, ACC, 6, NT>;
        case 8: return vg_scan_kernel<VT, ACC, 8, NT>;
    }
    return nullptr;
}

template <int VT, bool NT>
static scan_fn_t pick_acc(int acc, int U) {
    switch (acc) {
        case A_L2: return pick_u<VT, A_L2, NT>(U);
        case A_COS: return pick_u<VT, A_COS, NT>(U);
        case A_DOT: return pick_u<VT, A_DOT, NT>(U);
        case A_L1: return pick_u<VT, A_L1, NT>(U);
        case A_COSN:
            if constexpr (VT == T_F16 || VT == T_BF16) return pick_u<VT, A_COSN, NT>(U);
            return nullptr;
    }
    return nullptr;
}

template <bool NT>
static scan_fn_t pick_type(int vtype, int acc, int U) {
    switch (vtype) {
        case VG_TYPE_F32: return pick_acc<T_F32, NT>(acc, U);
        case VG_TYPE_U8: return pick_acc<T_U8, NT>(acc, U);
        case VG_TYPE_I8: return pick_acc<T_I8, NT>(acc, U);
        case VG_TYPE_F16: return pick_acc<T_F16, NT>(acc, U);
        case VG_TYPE_BF16: return pick_acc<T_BF16, NT>(acc, U);
    }
    return nullptr;
}

template <int VT, bool NT>
static scan_fn_t pick_long_acc(int acc) {
    switch (acc) {
        case A_L2: return vg_scan_long_kernel<VT, A_L2, NT>;
        case A_COS: return vg_scan_long_kernel<VT, A_COS, NT>;
        case A_DOT: return vg_scan_long_kernel<VT, A_DOT, NT>;
        case A_L1: return vg_scan_long_kernel<VT, A_L1, NT>;
    }
    return nullptr;
}

template <bool NT>
static scan_fn_t pick_long_type(int vtype, int acc) {
    switch (vtype) {
        case VG_TYPE_F32: return pick_long_acc<T_F32, NT>(acc);
        case VG_TYPE_U8: return pick_long_acc<T_U8, NT>(acc);
        case VG_TYPE_I8: return pick_long_acc<T_I8, NT>(acc);
        case VG_TYPE_F16: return pick_long_acc<T_F16, NT>(acc);
        case VG_TYPE_BF16: return pick_long_acc<T_BF16, NT>(acc);
    }
    return nullptr;
}

static scan_fn_t pick_kernel(int vtype, int acc, const Shape &s, bool nt) {
    if (s.long_rows) return nt ? pick_long_type<true>(vtype, acc) : pick_long_type<false>(vtype, acc);
    return nt ? pick_type<true>(vtype, acc, s.U) : pick_type<false>(vtype, acc, s.U);
}

// stream with non-temporal loads once the corpus cannot live in the 256 MiB Infinity Cache anyway
static bool use_nt_loads(const vg_corpus *c, int64_t n_rows) {
    int force = vg_sw(SW_VG_NT, -1);
    if (force >= 0) return force != 0;
    return n_rows * c->stride > (256ll << 20);
}

// batch order of the top-k scan (ScanArgs.order); VG_SCAN_ORDER=0 / 1 forces one
static int scan_order_for(const vg_corpus *c, int64_t n_rows) {
    const int force = vg_sw(SW_VG_SCAN_ORDER, -1);
    if (force >= 0) return force ? 1 : 0;
    (void)c; (void)n_rows;
    return 0;
}

int vg_metric_to_acc(int metric) {
    switch (metric) {
        case VG_DIST_L2: case VG_DIST_SQUARED_L2: return A_L2;
        case VG_DIST_COSINE: return A_COS;
        case VG_DIST_DOT: return A_DOT;
        case VG_DIST_L1: return A_L1;
    }
    return -1;
}

static const char *type_tag(int t) {
    switch (t) { case VG_TYPE_F32: return "f32"; case VG_TYPE_F16: return "f16"; case VG_TYPE_BF16: return "bf16";
                 case VG_TYPE_U8: return "u8"; case VG_TYPE_I8: return "i8"; }
    return "?";
}
static const char *acc_tag(int a) {
    switch (a) { case A_L2: return "l2"; case A_COS: return "cos"; case A_DOT: return "dot"; case A_L1: return "l1"; case A_COSN: return "cosn"; }
    return "?";
}


extern "C" const char *vg_scan_kernel_name(vg_corpus *c, int metric) {
    if (!c) return "";
    Shape s;
    int acc = vg_metric_to_acc(metric);
    if (acc < 0 || !choose_shape(c->nch, c->vtype, acc, &s)) return "";
    if (acc == A_COS && (c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16) && !s.long_rows && vg_sw(SW_VG_HALF_COSN, 1)) acc = A_COSN;
    if (!s.long_rows && vg_scan_filter_name(c, metric, c->kernel_name, sizeof(c->kernel_name))) return c->kernel_name;
    snprintf(c->kernel_name, sizeof(c->kernel_name), "scan%s_%s_%s_u%d_lpr%d%s", s.long_rows ? "_long" : "",
             type_tag(c->vtype), acc_tag(acc), s.U, 1 << s.lpr_log2, use_nt_loads(c, c->n_rows) ? "_nt" : "");
    return c->kernel_name;
}



// Final reduction of the per-CU lists (nlists <= 256) to the k best: one workgroup, parallel rank-select
// (vg_lists.h).  out_keys receives k keys ascending, VG_EMPTY_KEY padded to 64.
#define VG_MERGE_THREADS 1024
__global__ __launch_bounds__(VG_MERGE_THREADS) void vg_merge_kernel(const uint64_t *cand, int nlists, int k,
                                                                    uint64_t *out_keys, const unsigned long long *mirror_src,
                                                                    unsigned long long *mirror_dst) {
    __shared__ __attribute__((aligned(16))) uint8_t scratch[VG_SEL_SCRATCH_BYTES];
    // (filter scans) the counters the kernel in front finished with, copied to their pinned mirror: saves the copy command
    if (mirror_dst && blockIdx.x == 0 && threadIdx.x < 3) mirror_dst[threadIdx.x] = mirror_src[threadIdx.x];
    // one workgroup per query (gridDim.x = 1 for the single-query scan, NQ for vg_scan_multi_kernel)
    vg_select_lists(cand + (long long)blockIdx.x * nlists * VG_WAVE, nlists, k, out_keys + (long long)blockIdx.x * VG_WAVE, scratch);
}

// merge launch shared with vg_multi.hip: one workgroup per query
int vg_launch_merge(const uint64_t *dev_cand, int nlists, int k, uint64_t *dev_out_keys, int nq, hipStream_t stream) {
    hipLaunchKernelGGL(vg_merge_kernel, dim3((unsigned)nq), dim3(VG_MERGE_THREADS), 0, stream, dev_cand, nlists, k, dev_out_keys,
                       (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
    return (int)hipGetLastError();
}

// What one scan launch covers (ScanPlan, vg_internal.h).  The filter scans (vg_scan_filter.h) live in vg_filter.hip: their
// kernels are a translation unit of their own (compile time); vg_launch_scan_filter returns -1 when a scan is not served.
static int launch_scan(vg_corpus *c, int metric, const uint8_t *dev_query, int k, uint64_t *dev_out_keys,
                       float *dev_out_dist, hipStream_t stream, const ScanPlan &plan = ScanPlan());

int vg_launch_plain_scan(vg_corpus *c, int metric, const uint8_t *dev_query, int k, uint64_t *dev_out_keys, hipStream_t stream,
                         const ScanPlan &plan) {
    return launch_scan(c, metric, dev_query, k, dev_out_keys, nullptr, stream, plan);
}
int vg_launch_merge_one(const uint64_t *dev_cand, int nlists, int k, uint64_t *dev_out_keys, hipStream_t stream,
                        const unsigned long long *mirror_src, unsigned long long *mirror_dst) {
    hipLaunchKernelGGL(vg_merge_kernel, dim3(1), dim3(VG_MERGE_THREADS), 0, stream, dev_cand, nlists, k, dev_out_keys, mirror_src, mirror_dst);
    return (int)hipGetLastError();
}
int vg_plain_scan_shape(const vg_corpus *c, int metric, VgShape *out) {
    choose_shape(c->nch, c->vtype, vg_metric_to_acc(metric), out);
    return 0;
}

// Launch the scan (+ merge in top-k mode) on `stream`.  dev_query holds nch*16 zero-padded bytes.
static int launch_scan(vg_corpus *c, int metric, const uint8_t *dev_query, int k, uint64_t *dev_out_keys,
                       float *dev_out_dist, hipStream_t stream, const ScanPlan &plan) {
    int acc = vg_metric_to_acc(metric);
    if (acc < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    const int64_t n_rows = (plan.n_rows >= 0) ? std::min<int64_t>(plan.n_rows, c->n_rows) : c->n_rows;
    if (plan.ref_emit) c->ref_prefix_rows = -1;              // (set by whichever launch below really emits)
    Shape s;
    choose_shape(c->nch, c->vtype, acc, &s);
    // tie_order = reference over a SMALL corpus: the whole corpus is the replay's prefix - one "top-k + store" launch of the EX kernel
    // leaves every distance behind and an empty candidate stream (no pre-pass, no second scan on a tie)
    const bool whole_prefix = plan.ref_emit && !dev_out_dist && plan.n_rows < 0 && k > 0 && k <= VG_MAX_FUSED_K && !s.long_rows &&
                              n_rows < VG_REF_EMIT_MIN_ROWS;
    if (!whole_prefix && plan.allow_filter && plan.n_rows < 0 && !dev_out_dist && k <= VG_MAX_FUSED_K) {
        int rcf = vg_launch_scan_filter(c, metric, dev_query, k, dev_out_keys, stream, plan.ref_emit, plan.final_out);
        if (rcf != -1) return rcf;
    }
    // f16 / bf16 cosine: the row norms come from a cached vector (computed once per appended row) instead of being
    // re-accumulated in f64 on every scan - the f64 chain is what bounds these kernels, not HBM
    if (acc == A_COS && (c->vtype == VG_TYPE_F16 || c->vtype == VG_TYPE_BF16) && !s.long_rows && vg_sw(SW_VG_HALF_COSN, 1)) {
        int rcn = vg_ensure_row_norms(c);
        if (rcn != VG_OK) return rcn;
        acc = A_COSN;
        if (stream != c->stream) {                       // the norm pass ran on the corpus stream
            if (!c->norm_ev) HIP_TRY(hipEventCreateWithFlags(&c->norm_ev, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c->norm_ev, c->stream));
            HIP_TRY(hipStreamWaitEvent(stream, c->norm_ev, 0));
        }
    }
    // tie_order = reference: the prefix pass (this kernel over the first P rows: top-k + store) runs in front of an emitting scan; its
    // k-th best is every list's start threshold and the line below which an accepted row is emitted.  Both are EX kernels (vg_scan_ex.hip).
    const bool emitting = plan.ref_emit && !dev_out_dist && !s.long_rows && plan.n_rows < 0 && k <= VG_MAX_FUSED_K &&
                          n_rows >= VG_REF_EMIT_MIN_ROWS;
    float *store_prefix = plan.store_prefix;
    unsigned long long *emit_reset = plan.emit_reset;
    if (whole_prefix) {
        int rcb = vg_ensure_ref_buffers(c, n_rows);
        if (rcb != VG_OK) return rcb;
        store_prefix = c->d_ref_prefix;
        emit_reset = c->d_below;
        c->ref_prefix_rows = n_rows;
    }
    const bool prefix_pass = store_prefix && !dev_out_dist && !s.long_rows && k > 0;
    scan_fn_t fn = (emitting || prefix_pass) ? vg_pick_scan_kernel_ex(c->vtype, acc, s.U) : pick_kernel(c->vtype, acc, s, use_nt_loads(c, n_rows));
    if (!fn) return vg_fail(VG_ERR_UNSUPPORTED, "no scan kernel for type %s", type_tag(c->vtype));

    const int rpb = VG_WAVE >> s.lpr_log2;
    const long long nbatch = (n_rows + rpb - 1) / rpb;
    // 16-wave workgroups: one per CU is what ~96 VGPRs admit (5 waves/SIMD); a second one only queues behind it
    const int bpc = std::max(1, std::min(8, vg_sw(SW_VG_BLOCKS_PER_CU, 1)));
    long long blocks = (nbatch + VG_WAVES_PER_BLOCK - 1) / VG_WAVES_PER_BLOCK;
    blocks = std::max<long long>(1, std::min<long long>(blocks, (long long)c->cu_count * bpc));
    blocks = std::min<long long>(blocks, VG_SEL_MAX_HEADS);          // the final rank-select handles <= 256 lists
    // small corpora: at least two batches per wavefront - the merge kernel's time grows with the number of per-CU lists (6.6 us for
    // 32, 14 us for 256) and is most of a 10k-row query; round 3's wider shapes halved the rows per batch, i.e. doubled the lists
    blocks = std::min<long long>(blocks, std::max<long long>(32, nbatch / (2 * VG_WAVES_PER_BLOCK)));

    ScanArgs a{};
    a.rows = c->d_rows;
    a.query = dev_query;
    a.cand = plan.lists_out ? plan.lists_out : c->d_cand;
    a.out_dist = dev_out_dist;
    a.n_rows = n_rows;
    a.stride = c->stride;
    a.nch = c->nch;
    a.lpr_log2 = s.lpr_log2;
    a.k = k;
    a.root = (metric == VG_DIST_L2) ? 1 : 0;
    a.dim = c->dim;
    a.row_nn = (acc == A_COSN) ? c->d_xnorm : nullptr;
    a.init_keys = nullptr;
    a.emit = nullptr;
    a.emit_cap = 0;
    a.emit_reset = emit_reset;
    a.order = scan_order_for(c, n_rows);
    if (prefix_pass) a.out_dist = store_prefix;            // top-k + store (EX kernel, k > 0)
    else a.emit_reset = nullptr;
    size_t qbytes = (size_t)c->nch * 16;
    if (s.long_rows) {
        const size_t slice = (size_t)VG_WAVE * VG_LONG_U;               // the long kernel pads the query to whole slices
        qbytes = ((c->nch + slice - 1) / slice) * slice * 16;
    }
    size_t smem = std::max<size_t>(qbytes, (size_t)VG_PUBLISH_LDS_BYTES);
    a.store_lds_off = 0;
    if (dev_out_dist && !s.long_rows) {             // store mode: a staging area per wavefront behind the query
        a.store_lds_off = (int)((qbytes + 255) / 256 * 256);
        smem = std::max<size_t>(smem, (size_t)a.store_lds_off + (size_t)VG_WAVES_PER_BLOCK * VG_STORE_FLOATS * sizeof(float));
    }

    // host appends are only enqueued on the corpus stream: a scan on ANOTHER stream must wait for them
    if (c->append_pending && stream != c->stream) HIP_TRY(hipStreamWaitEvent(stream, c->append_ev, 0));

    hipEvent_t *evs = plan.record ? vg_prof_slot(c, (uint8_t)((dev_out_dist == nullptr ? VG_EVF_MERGE : 0) | (emitting ? VG_EVF_PREPASS : 0))) : nullptr;
    if (evs) hipEventRecord(evs[0], stream);
    if (emitting) {
        const int64_t P = vg_ref_prefix_for(n_rows);
        int rcb = vg_ensure_ref_buffers(c, P);
        if (rcb != VG_OK) return rcb;
        ScanPlan pre;
        pre.n_rows = P;
        pre.allow_filter = false;
        pre.record = false;
        pre.store_prefix = c->d_ref_prefix;
        pre.emit_reset = c->d_below;
        int rcp = launch_scan(c, metric, dev_query, k, dev_out_keys, nullptr, stream, pre);
        if (rcp != VG_OK) return rcp;
        a.init_keys = dev_out_keys;                  // read by every workgroup before the final merge overwrites it
        a.emit = c->d_below;
        a.emit_cap = VG_BELOW_CAP;
        c->ref_prefix_rows = P;
        if (evs) hipEventRecord(evs[1], stream);
    }
    if (smem > 64 * 1024)          // very long rows: the query alone needs more than the default dynamic-LDS window
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(VG_BLOCK), smem, stream, a);
    if (evs) hipEventRecord(evs[2], stream);
    if (plan.lists_out) {                                   // the caller's kernel reads the lists itself: no merge launch
        if (plan.n_lists_out) *plan.n_lists_out = (int)blocks;
    } else if (!dev_out_dist) {
        hipLaunchKernelGGL(vg_merge_kernel, dim3(1), dim3(VG_MERGE_THREADS), 0, stream,
                           (const uint64_t *)c->d_cand, (int)blocks, k, plan.final_out ? plan.final_out : dev_out_keys,
                           (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
    }
    if (evs) hipEventRecord(evs[3], stream);
    HIP_TRY(hipGetLastError());
    return VG_OK;
}

// latency path for small corpora (see vg_scan_topk): below this size the H2D/D2H staging copies dominate
static bool host_direct(const vg_corpus *c) {
    const int v = vg_sw(SW_VG_HOST_DIRECT, -1);
    if (v >= 0) return v != 0;
    return c->n_rows * c->stride <= (64ll << 20);
}

static void stage_query(vg_corpus *c, const void *query) {
    memset(c->h_query, 0, (size_t)c->stride);
    memcpy(c->h_query, query, (size_t)c->dim * c->es);
}

// kernel times of ring slot `slot` (waits for that launch to finish)
// scan_ms is ONE kernel (the scan / filter-scan kernel); a filter scan's plain-f32 pre-pass (its scan + merge) is prepass_ms
static void slot_times(vg_corpus *c, int slot, float *scan_ms, float *merge_ms, float *prepass_ms) {
    hipEvent_t *evs = &c->ev[(size_t)slot * VG_PROF_EVS];
    const uint8_t fl = c->ev_flags[(size_t)slot];
    hipEventSynchronize(evs[3]);
    *scan_ms = 0.f; *merge_ms = 0.f; *prepass_ms = 0.f;
    if (fl & VG_EVF_PREPASS) hipEventElapsedTime(prepass_ms, evs[0], evs[1]);
    hipEventElapsedTime(scan_ms, evs[(fl & VG_EVF_PREPASS) ? 1 : 0], evs[2]);
    if (fl & VG_EVF_MERGE) hipEventElapsedTime(merge_ms, evs[2], evs[3]);
}

void vg_collect_timing(vg_corpus *c) {
    if (!c->profiling || c->prof_launches == 0) return;
    slot_times(c, (int)((c->prof_launches - 1) % VG_PROF_RING), &c->last_scan_ms, &c->last_merge_ms, &c->last_prepass_ms);
}

extern "C" float vg_key_distance(uint64_t key) { return vg_sortable_f32((uint32_t)(key >> 32)); }
extern "C" uint32_t vg_key_position(uint64_t key) { return (uint32_t)(key & 0xFFFFFFFFull); }

extern "C" int vg_scan_topk_device(vg_corpus *c, int metric, const void *dev_query, int k, uint64_t *dev_out_keys,
                                   void *stream) {
    if (!c || !dev_query || !dev_out_keys) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_device: NULL argument");
    if (k < 1 || k > VG_MAX_FUSED_K) return vg_fail(VG_ERR_UNSUPPORTED, "vg_scan_topk_device: k must be in 1..%d", VG_MAX_FUSED_K);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (c->n_rows == 0) {
        HIP_TRY(hipMemsetAsync(dev_out_keys, 0xFF, VG_WAVE * sizeof(uint64_t), st));
        return VG_OK;
    }
    return launch_scan(c, metric, (const uint8_t *)dev_query, k, dev_out_keys, nullptr, st);
}

extern "C" int vg_scan_distances_device(vg_corpus *c, int metric, const void *dev_query, float *dev_out_dist, void *stream) {
    if (!c || !dev_query || !dev_out_dist) return vg_fail(VG_ERR_INVALID, "vg_scan_distances_device: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_rows == 0) return VG_OK;
    return launch_scan(c, metric, (const uint8_t *)dev_query, 0, nullptr, dev_out_dist, stream ? (hipStream_t)stream : c->stream);
}

static int ensure_dist_buffer(vg_corpus *c) {
    c->dist_valid_rows = 0;
    if (c->d_dist_cap >= c->n_rows) return VG_OK;
    if (c->d_dist) { hipFree(c->d_dist); c->d_dist = nullptr; c->d_dist_cap = 0; }
    HIP_TRY(hipMalloc(&c->d_dist, (size_t)c->n_rows * sizeof(float)));
    c->d_dist_cap = c->n_rows;
    return VG_OK;
}

extern "C" int vg_scan_distances(vg_corpus *c, int metric, const void *query, float *out_dist_host) {
    if (!c || !query || !out_dist_host) return vg_fail(VG_ERR_INVALID, "vg_scan_distances: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_rows == 0) return VG_OK;
    int rc = ensure_dist_buffer(c);
    if (rc != VG_OK) return rc;
    stage_query(c, query);
    HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
    rc = launch_scan(c, metric, c->d_query, 0, nullptr, c->d_dist, c->stream);
    if (rc != VG_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_dist_host, c->d_dist, (size_t)c->n_rows * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    vg_collect_timing(c);
    return VG_OK;
}

// All N distances of one query, left in the corpus' device buffer (enqueued only - the readers below synchronise):
// the first half of the tie_order = reference scan (vg_reforder.hip) and of its multi-device form (vg_shards.hip).
extern "C" int vg_scan_distances_resident(vg_corpus *c, int metric, const void *query) {
    if (!c || !query) return vg_fail(VG_ERR_INVALID, "vg_scan_distances_resident: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_rows == 0) { c->dist_valid_rows = 0; return VG_OK; }
    int rc = ensure_dist_buffer(c);
    if (rc != VG_OK) return rc;
    stage_query(c, query);
    HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
    rc = launch_scan(c, metric, c->d_query, 0, nullptr, c->d_dist, c->stream);
    if (rc != VG_OK) return rc;
    c->dist_valid_rows = c->n_rows;
    return VG_OK;
}

// k > 64 (beyond the fused one-slot-per-lane list): store-mode scan -> key build -> device radix sort
// (vg_select.hip).  Nothing is computed or ordered on the host; k keys come back and are decoded.
extern "C" int vg_select_temp_bytes(long long n, size_t *bytes);
extern "C" int vg_select_sorted_keys(const float *dist, long long n, uint64_t *keys_tmp, uint64_t *keys_sorted,
                                     void *temp, size_t temp_bytes, hipStream_t stream);

extern "C" int vg_select_topk_keys(const float *dist, long long n, uint32_t k, uint64_t *keys_tmp, uint64_t *keys_sorted,
                                   uint32_t cap, void *temp, size_t temp_bytes, uint32_t *state, hipStream_t stream,
                                   uint32_t *out_count);

static int scan_topk_large_k(vg_corpus *c, int metric, const void *query, int k, uint64_t *out_keys, int *out_count) {
    int rc = ensure_dist_buffer(c);
    if (rc != VG_OK) return rc;
    if (c->sel_cap < c->n_rows) {
        if (c->d_sel_keys) hipFree(c->d_sel_keys);
        if (c->d_sel_sorted) hipFree(c->d_sel_sorted);
        if (c->d_sel_temp) hipFree(c->d_sel_temp);
        c->d_sel_keys = c->d_sel_sorted = nullptr; c->d_sel_temp = nullptr; c->sel_cap = 0;
        if (vg_select_temp_bytes(c->n_rows, &c->sel_temp_bytes) != 0) return vg_fail(VG_ERR_HIP, "radix sort temp-size query failed");
        HIP_TRY(hipMalloc(&c->d_sel_keys, (size_t)c->n_rows * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&c->d_sel_sorted, (size_t)c->n_rows * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&c->d_sel_temp, c->sel_temp_bytes ? c->sel_temp_bytes : 16));
        c->sel_cap = c->n_rows;
    }
    stage_query(c, query);
    HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
    rc = launch_scan(c, metric, c->d_query, 0, nullptr, c->d_dist, c->stream);
    if (rc != VG_OK) return rc;
    size_t take = (size_t)std::min<int64_t>((int64_t)k, c->n_rows);
    // radix select (three histogram passes + gather + a sort of ~k keys); the full N-key sort only when the k-th
    // distance has so many ties that the gathered set would not fit
    int sel = 1;
    if (vg_sw(SW_VG_RADIX_SELECT, 1)) {
        if (!c->d_sel_state) HIP_TRY(hipMalloc(&c->d_sel_state, (4 + 2048) * sizeof(uint32_t)));
        uint32_t got = 0;
        sel = vg_select_topk_keys(c->d_dist, c->n_rows, (uint32_t)take, c->d_sel_keys, c->d_sel_sorted, (uint32_t)std::min<int64_t>(c->sel_cap, 0xFFFFFFFFll),
                                  c->d_sel_temp, c->sel_temp_bytes, c->d_sel_state, c->stream, &got);
        if (sel > 1) return vg_fail(VG_ERR_HIP, "device radix select failed: %s", hipGetErrorString((hipError_t)sel));
        if (sel == 0) take = got;
    }
    if (sel != 0 && vg_select_sorted_keys(c->d_dist, c->n_rows, c->d_sel_keys, c->d_sel_sorted, c->d_sel_temp, c->sel_temp_bytes, c->stream) != 0)
        return vg_fail(VG_ERR_HIP, "device key sort failed: %s", hipGetErrorString(hipGetLastError()));
    if (take) HIP_TRY(hipMemcpyAsync(out_keys, c->d_sel_sorted, take * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    vg_collect_timing(c);
    int cnt = 0;
    while ((size_t)cnt < take && out_keys[cnt] != VG_EMPTY_KEY) ++cnt;     // NaN / +Inf rows sort last and are not results
    *out_count = cnt;
    return VG_OK;
}

// fused path (k <= 64), split in two so that a caller can put several shards in flight before waiting for any:
// enqueue = stage the query + launch scan and merge (+ copy the 64 keys back); collect = wait + hand out the keys
int vg_scan_topk_enqueue_plan(vg_corpus *c, int metric, const void *query, int k, bool ref_emit) {
    if (!c || !query) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_enqueue: NULL argument");
    if (k < 1 || k > VG_MAX_FUSED_K) return vg_fail(VG_ERR_UNSUPPORTED, "vg_scan_topk_enqueue: k must be in 1..%d", VG_MAX_FUSED_K);
    if (vg_metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    c->enqueued = false;
    if (c->n_rows == 0) return VG_OK;
    HIP_TRY(hipSetDevice(c->device));
    stage_query(c, query);
    ScanPlan plan;
    plan.ref_emit = ref_emit;
    int rc;
    if (host_direct(c)) {
        // small corpus: the two staging copies cost more than the scan.  h_query / h_keys are pinned, device-mapped
        // host buffers: the kernels read the query and write the k winners straight across the host link.
        rc = launch_scan(c, metric, c->h_query, k, c->h_keys, nullptr, c->stream, plan);
        if (rc != VG_OK) return rc;
    } else {
        // the query is staged into HBM (every workgroup reads it); the k winners are written by the final merge's one workgroup
        // straight into the pinned h_keys (VG_KEYS_DIRECT=0: into d_keys + a copy command, the form measured against it)
        const bool keys_direct = vg_sw(SW_VG_KEYS_DIRECT, 1) != 0;
        HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
        if (keys_direct) plan.final_out = c->h_keys;
        rc = launch_scan(c, metric, c->d_query, k, c->d_keys, nullptr, c->stream, plan);
        if (rc != VG_OK) return rc;
        if (!keys_direct) HIP_TRY(hipMemcpyAsync(c->h_keys, c->d_keys, VG_WAVE * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    }
    c->enqueued = true;
    return VG_OK;
}
extern "C" int vg_scan_topk_enqueue(vg_corpus *c, int metric, const void *query, int k) {
    return vg_scan_topk_enqueue_plan(c, metric, query, k, false);
}
// the same, but the 64 keys stay in c->d_keys on the device (no host-direct path, no copy back): what a device-side gather of the
// shards' candidates reads (vg_shards.hip, RCCL form).  An empty corpus leaves 64 EMPTY keys there.
int vg_scan_topk_enqueue_dev(vg_corpus *c, int metric, const void *query, int k) {
    if (!c || !query) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_enqueue_dev: NULL argument");
    if (k < 1 || k > VG_MAX_FUSED_K) return vg_fail(VG_ERR_UNSUPPORTED, "vg_scan_topk_enqueue_dev: k must be in 1..%d", VG_MAX_FUSED_K);
    if (vg_metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    HIP_TRY(hipSetDevice(c->device));
    c->enqueued = false;
    if (c->n_rows == 0) {
        HIP_TRY(hipMemsetAsync(c->d_keys, 0xFF, VG_WAVE * sizeof(uint64_t), c->stream));
        return VG_OK;
    }
    stage_query(c, query);
    HIP_TRY(hipMemcpyAsync(c->d_query, c->h_query, (size_t)c->stride, hipMemcpyHostToDevice, c->stream));
    return launch_scan(c, metric, c->d_query, k, c->d_keys, nullptr, c->stream);
}

extern "C" int vg_scan_topk_collect(vg_corpus *c, uint64_t *out_keys64) {
    if (!c || !out_keys64) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_collect: NULL argument");
    if (!c->enqueued) {                                        // empty corpus (or nothing enqueued): no candidates
        for (int i = 0; i < VG_WAVE; ++i) out_keys64[i] = VG_EMPTY_KEY;
        return VG_OK;
    }
    c->enqueued = false;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    vg_collect_timing(c);
    memcpy(out_keys64, c->h_keys, VG_WAVE * sizeof(uint64_t));
    return VG_OK;
}

extern "C" int vg_scan_topk_keys(vg_corpus *c, int metric, const void *query, int k, uint64_t *out_keys, int *out_count) {
    if (!c || !query || !out_count) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_keys: NULL argument");
    *out_count = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_keys) return vg_fail(VG_ERR_INVALID, "vg_scan_topk_keys: NULL output");
    if (vg_metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    HIP_TRY(hipSetDevice(c->device));
    if (k > VG_MAX_FUSED_K) return scan_topk_large_k(c, metric, query, k, out_keys, out_count);
    uint64_t keys[VG_WAVE];
    int rc = vg_scan_topk_enqueue(c, metric, query, k);
    if (rc == VG_OK) rc = vg_scan_topk_collect(c, keys);
    if (rc != VG_OK) return rc;
    int cnt = 0;
    while (cnt < k && keys[cnt] != VG_EMPTY_KEY) { out_keys[cnt] = keys[cnt]; ++cnt; }
    *out_count = cnt;
    return VG_OK;
}

extern "C" int vg_scan_topk(vg_corpus *c, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist,
                            int *out_count) {
    if (!c || !query || !out_count) return vg_fail(VG_ERR_INVALID, "vg_scan_topk: NULL argument");
    *out_count = 0;
    if (k <= 0 || c->n_rows == 0) return VG_OK;
    if (!out_rowids || !out_dist) return vg_fail(VG_ERR_INVALID, "vg_scan_topk: NULL output");
    if (c->tie_order == VG_TIE_REFERENCE) return vg_scan_topk_reference(c, metric, query, k, out_rowids, out_dist, out_count);
    std::vector<uint64_t> keys((size_t)std::min<int64_t>((int64_t)k, c->n_rows));
    int cnt = 0;
    int rc = vg_scan_topk_keys(c, metric, query, (int)keys.size(), keys.data(), &cnt);
    if (rc != VG_OK) return rc;
    for (int i = 0; i < cnt; ++i) {
        out_dist[i] = (double)vg_key_distance(keys[(size_t)i]);
        out_rowids[i] = vg_corpus_rowid_at(c, (int64_t)vg_key_position(keys[(size_t)i]));
    }
    *out_count = cnt;
    return VG_OK;
}

extern "C" int vg_rownorm_launch(const float *dev_rows, long long row0, long long n, long long stride_bytes, float *dev_out,
                                 hipStream_t stream);

// Row norms, computed once per appended row and kept next to the corpus.  f32 corpora: ||row|| (batched cosine / L2);
// f16 / bf16 corpora: (float) sum x^2 (single-query cosine, A_COSN).
int vg_ensure_row_norms(vg_corpus *c) {
    if (c->xnorm_cap < c->n_rows) {
        float *nb = nullptr;
        const int64_t cap = std::max<int64_t>(c->cap_rows, c->n_rows);
        HIP_TRY(hipMalloc(&nb, (size_t)(cap + 192) * sizeof(float)));     // + slack: the batch kernels fetch the norms of four whole tiles at a time
        HIP_TRY(hipMemsetAsync(nb, 0, (size_t)(cap + 192) * sizeof(float), c->stream));
        if (c->d_xnorm && c->xnorm_rows > 0)
            HIP_TRY(hipMemcpyAsync(nb, c->d_xnorm, (size_t)c->xnorm_rows * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        if (c->d_xnorm) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->d_xnorm); }
        c->d_xnorm = nb;
        c->xnorm_cap = cap;
    }
    if (c->xnorm_rows < c->n_rows) {
        const long long n = c->n_rows - c->xnorm_rows;
        int rc = 0;
        if (c->vtype == VG_TYPE_F32) {
            rc = vg_rownorm_launch((const float *)c->d_rows, c->xnorm_rows, n, c->stride, c->d_xnorm, c->stream);
        } else {                                   // f16 / bf16: (float) sum x^2, vg_half_rownorm_kernel (vg_scan.h)
            long long blocks = std::min<long long>((n * 16 + 255) / 256, 256 * 32);
            if (c->vtype == VG_TYPE_F16)
                hipLaunchKernelGGL((vg_half_rownorm_kernel<T_F16>), dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_rows,
                                   (long long)c->xnorm_rows, n, (long long)c->stride, c->nch, c->d_xnorm);
            else
                hipLaunchKernelGGL((vg_half_rownorm_kernel<T_BF16>), dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_rows,
                                   (long long)c->xnorm_rows, n, (long long)c->stride, c->nch, c->d_xnorm);
            rc = (int)hipGetLastError();
        }
        if (rc != 0) return vg_fail(VG_ERR_HIP, "row-norm pass failed: %s", hipGetErrorString((hipError_t)rc));
        c->xnorm_rows = c->n_rows;
    }
    return VG_OK;
}

// ------------------------------------------------------------------------------------------------ instrumentation

extern "C" int vg_set_profiling(vg_corpus *c, int enabled) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (enabled && c->ev.empty()) {
        c->ev.assign((size_t)VG_PROF_RING * VG_PROF_EVS, nullptr);
        c->ev_flags.assign((size_t)VG_PROF_RING, 0);
        for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    }
    c->profiling = enabled != 0;
    c->prof_launches = 0;
    return VG_OK;
}

extern "C" int vg_profile_mean_ms_ex(vg_corpus *c, int *n_launches, float *scan_ms, float *merge_ms, float *prepass_ms) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    long long n = std::min<long long>(c->prof_launches, VG_PROF_RING);
    double s = 0.0, m = 0.0, p = 0.0;
    for (long long i = 0; i < n; ++i) {
        float a = 0.f, b = 0.f, pp = 0.f;
        slot_times(c, (int)((c->prof_launches - 1 - i) % VG_PROF_RING), &a, &b, &pp);
        s += a; m += b; p += pp;
    }
    if (n_launches) *n_launches = (int)n;
    if (scan_ms) *scan_ms = n ? (float)(s / n) : 0.f;
    if (merge_ms) *merge_ms = n ? (float)(m / n) : 0.f;
    if (prepass_ms) *prepass_ms = n ? (float)(p / n) : 0.f;
    return VG_OK;
}

extern "C" int vg_profile_mean_ms(vg_corpus *c, int *n_launches, float *scan_ms, float *merge_ms) {
    return vg_profile_mean_ms_ex(c, n_launches, scan_ms, merge_ms, nullptr);
}

// Filter scan instrumentation: rows evaluated exactly by the filter-scan launches since the last call (waits for the corpus
// stream, so that the launches enqueued on it are counted).
extern "C" int vg_filter_exact_evals(vg_corpus *c, unsigned long long *out_evals) {
    if (!c || !out_evals) return vg_fail(VG_ERR_INVALID, "vg_filter_exact_evals: NULL argument");
    *out_evals = 0;
    if (!c->d_filter_evals) return VG_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long now = 0;
    HIP_TRY(hipMemcpy(&now, c->d_filter_evals, sizeof(now), hipMemcpyDeviceToHost));
    *out_evals = now - c->filter_evals_read;
    c->filter_evals_read = now;
    return VG_OK;
}

extern "C" int vg_filter_guard_cooldown(vg_corpus *c) { return c ? c->filter_cooldown : 0; }

// Per-corpus switch of the filter scan (the extension's scan_filter= option): 0 = plain f32 scans, 1 = filter scan where it
// serves, -1 = default (the VG_SCAN_FILTER environment switch, else on).  Turning it off releases nothing by itself.
extern "C" int vg_corpus_set_scan_filter(vg_corpus *c, int mode) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    c->scan_filter_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return VG_OK;
}

extern "C" int vg_last_kernel_ms(vg_corpus *c, float *scan_ms, float *merge_ms) {
    if (!c) return vg_fail(VG_ERR_INVALID, "corpus is NULL");
    vg_collect_timing(c);      // waits for the recorded events of the last (possibly still running) scan
    if (scan_ms) *scan_ms = c->last_scan_ms;
    if (merge_ms) *merge_ms = c->last_merge_ms;
    return VG_OK;
}