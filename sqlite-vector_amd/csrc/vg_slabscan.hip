// vg_slabscan.hip - out-of-core scans: one query over a table that is handed over slab by slab (include/vectorgpu.h: vg_slab_scan_*).
//
// The reference has no residency to lose: every vector_full_scan walks the table's rows through one statement and offers each row to k
// slots (sqlite-vector.c:2071-2113; the quantized scan walks the preloaded buffer or the chunks, :2121-2157 / :2193-2234).  A table
// that fits the device is staged once and scanned where it lies (vg_corpus / vg_shards); a table that does not fit is scanned the
// reference's way, with the device as the distance engine: TWO slab corpora, the caller's rows fill one (pinned bounce buffers -> HBM,
// vg_corpus_append) while a host thread scans the other, and the state that travels from slab to slab is the reference's own state -
//   tie_order = position   the k best (distance, rowid) so far; a slab's own top-k (ordered by (distance, position)) is merged behind
//                          them, earlier slabs first among equal distances: the order of one corpus holding all rows;
//   tie_order = reference  the k slots with their history (VgRefSlots).  The first slab is replayed like a corpus (prefix + the rows
//                          below the bound); every later slab offers only its rows below the bound reached - a handful;
//   k = 0                  every distance and rowid, appended slab by slab (the *_stream functions).
// The slab corpora never build derived data (filter scans off: each slab's rows are seen once).
#include "vg_internal.h"
#include "vg_refslots.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct vg_slab_scan {
    int device = 0, vtype = 0, dim = 0, metric = 0, k = 0, tie_order = 0, es = 1;
    int64_t slab_rows = 0, rowid_base = 1;
    std::vector<uint8_t> query;
    vg_corpus *slab[2] = {nullptr, nullptr};
    int64_t base[2] = {0, 0};                  // global scan position of each slab's first row
    int fill = 0;                              // the slab the caller's rows go to
    int64_t total = 0;                         // rows handed over so far
    int slabs_done = 0;
    std::thread worker;                        // scans the other slab
    bool busy = false;
    int worker_rc = VG_OK;
    std::string worker_msg;
    bool finished = false;
    // position order: the k best so far, ascending by (distance, scan position)
    std::vector<double> best_d;
    std::vector<int64_t> best_id;
    // reference order: the slots and the rowid of each slot's occupant
    VgRefSlots slots;
    std::vector<int64_t> slot_rowid;
    // k = 0
    std::vector<float> all_d;
    std::vector<int64_t> all_id;
};

static int slab_scan_one(vg_slab_scan *s, int which) {
    vg_corpus *c = s->slab[which];
    const int64_t n = vg_corpus_rows(c), g0 = s->base[which];
    if (n == 0) return VG_OK;
    const void *q = s->query.data();
    int rc;
    if (s->k == 0) {
        const size_t o = s->all_d.size();
        s->all_d.resize(o + (size_t)n);
        s->all_id.resize(o + (size_t)n);
        if ((rc = vg_scan_distances(c, s->metric, q, s->all_d.data() + o)) != VG_OK) return rc;
        for (int64_t i = 0; i < n; ++i) s->all_id[o + (size_t)i] = vg_corpus_rowid_at(c, i);
    } else if (s->tie_order == VG_TIE_REFERENCE) {
        if ((rc = vg_ref_replay_slab(c, s->metric, q, s->k, s->slots, g0, s->slabs_done == 0)) != VG_OK) return rc;
        for (int i = 0; i < s->k; ++i)                                           // the occupants this slab brought: their rowids, while the slab is here
            if (s->slots.pos[(size_t)i] >= g0) s->slot_rowid[(size_t)i] = vg_corpus_rowid_at(c, s->slots.pos[(size_t)i] - g0);
    } else {
        std::vector<int64_t> ids((size_t)s->k);
        std::vector<double> d((size_t)s->k);
        int cnt = 0;
        if ((rc = vg_scan_topk(c, s->metric, q, s->k, ids.data(), d.data(), &cnt)) != VG_OK) return rc;
        std::vector<double> md;
        std::vector<int64_t> mi;
        md.reserve((size_t)s->k);
        mi.reserve((size_t)s->k);
        size_t a = 0;
        int b = 0;
        while ((int)md.size() < s->k && (a < s->best_d.size() || b < cnt)) {
            // equal distances: the earlier slab's row lies earlier in the scan
            const bool old = a < s->best_d.size() && (b >= cnt || !(d[(size_t)b] < s->best_d[a]));
            if (old) { md.push_back(s->best_d[a]); mi.push_back(s->best_id[a]); ++a; }
            else { md.push_back(d[(size_t)b]); mi.push_back(ids[(size_t)b]); ++b; }
        }
        s->best_d.swap(md);
        s->best_id.swap(mi);
    }
    ++s->slabs_done;
    return VG_OK;
}

static int slab_join(vg_slab_scan *s) {
    if (!s->busy) return VG_OK;
    s->worker.join();
    s->busy = false;
    if (s->worker_rc != VG_OK) return vg_fail(s->worker_rc, "%s", s->worker_msg.c_str());
    return VG_OK;
}

// the filled slab goes to the scanning thread (after the scan in flight - of the OTHER slab - has ended: the state is sequential), and the
// other slab, whose rows nobody needs any more, takes the next rows
static int slab_dispatch(vg_slab_scan *s) {
    int rc = slab_join(s);
    if (rc != VG_OK) return rc;
    const int which = s->fill;
    s->busy = true;
    s->worker_rc = VG_OK;
    s->worker = std::thread([s, which] {
        hipSetDevice(s->device);
        const int r = slab_scan_one(s, which);
        if (r != VG_OK) { s->worker_rc = r; s->worker_msg = vg_last_error(); }
    });
    s->fill ^= 1;
    if ((rc = vg_corpus_clear(s->slab[s->fill])) != VG_OK) return rc;
    s->base[s->fill] = s->total;
    return vg_corpus_set_rowid_base(s->slab[s->fill], s->rowid_base + s->total);
}

extern "C" int vg_slab_scan_begin(int device, int vtype, int dim, int metric, const void *query, int k, int tie_order, int64_t slab_rows,
                                  int64_t rowid_base, vg_slab_scan **out) {
    if (!out) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_begin: out is NULL");
    *out = nullptr;
    if (!query || k < 0 || slab_rows < 1) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_begin: query, k >= 0 and slab_rows >= 1 are needed");
    if (vg_metric_to_acc(metric) < 0) return vg_fail(VG_ERR_INVALID, "unknown distance metric %d", metric);
    vg_slab_scan *s = new vg_slab_scan();
    s->device = device; s->vtype = vtype; s->dim = dim; s->metric = metric; s->k = k; s->tie_order = tie_order;
    s->slab_rows = slab_rows; s->rowid_base = rowid_base;
    s->es = (vtype == VG_TYPE_F32) ? 4 : (vtype == VG_TYPE_F16 || vtype == VG_TYPE_BF16) ? 2 : 1;
    for (int i = 0; i < 2; ++i) {
        int rc = vg_corpus_create(device, vtype, dim, slab_rows, &s->slab[i]);
        if (rc == VG_OK) rc = vg_corpus_reserve(s->slab[i], slab_rows);
        if (rc == VG_OK) rc = vg_corpus_set_scan_filter(s->slab[i], 0);
        if (rc == VG_OK) rc = vg_corpus_set_tie_order(s->slab[i], VG_TIE_POSITION);
        if (rc == VG_OK) rc = vg_corpus_set_rowid_base(s->slab[i], rowid_base);
        if (rc != VG_OK) { vg_slab_scan_destroy(s); return rc; }
    }
    s->query.assign((const uint8_t *)query, (const uint8_t *)query + (size_t)dim * (size_t)s->es);
    if (k > 0 && tie_order == VG_TIE_REFERENCE) { s->slots.init(k); s->slot_rowid.assign((size_t)k, 0); }
    *out = s;
    return VG_OK;
}

template <typename Put>
static int slab_feed(vg_slab_scan *s, int64_t n, Put put) {
    if (s->finished) return vg_fail(VG_ERR_INVALID, "vg_slab_scan: rows after finish");
    int64_t done = 0;
    while (done < n) {
        vg_corpus *c = s->slab[s->fill];
        const int64_t take = std::min<int64_t>(n - done, s->slab_rows - vg_corpus_rows(c));
        int rc = put(c, done, take);
        if (rc != VG_OK) return rc;
        done += take;
        s->total += take;
        if (vg_corpus_rows(c) >= s->slab_rows && (rc = slab_dispatch(s)) != VG_OK) return rc;
    }
    return VG_OK;
}

extern "C" int vg_slab_scan_rows(vg_slab_scan *s, const void *host_rows, int64_t n_rows, int64_t row_stride_bytes, const int64_t *rowids) {
    if (!s || (!host_rows && n_rows > 0) || n_rows < 0) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_rows: bad argument");
    const int64_t stride = row_stride_bytes > 0 ? row_stride_bytes : (int64_t)s->dim * s->es;
    return slab_feed(s, n_rows, [&](vg_corpus *c, int64_t off, int64_t take) {
        return vg_corpus_append(c, (const uint8_t *)host_rows + off * stride, take, stride, rowids ? rowids + off : nullptr);
    });
}

extern "C" int vg_slab_scan_records(vg_slab_scan *s, const void *host_records, int64_t n_records) {
    if (!s || (!host_records && n_records > 0) || n_records < 0) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_records: bad argument");
    const int64_t rec = 8 + (int64_t)s->dim * s->es;
    return slab_feed(s, n_records, [&](vg_corpus *c, int64_t off, int64_t take) {
        return vg_corpus_append_records(c, (const uint8_t *)host_records + off * rec, take);
    });
}

extern "C" int vg_slab_scan_finish(vg_slab_scan *s, int64_t *out_rowids, double *out_dist, int *out_count) {
    if (!s || !out_count) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_finish: NULL argument");
    *out_count = 0;
    if (!s->finished) {
        int rc = VG_OK;
        if (vg_corpus_rows(s->slab[s->fill]) > 0) rc = slab_dispatch(s);     // the ragged last slab
        const int rc2 = slab_join(s);
        s->finished = true;
        if (rc != VG_OK) return rc;
        if (rc2 != VG_OK) return rc2;
    }
    if (s->k == 0) return VG_OK;
    if (!out_rowids || !out_dist) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_finish: NULL output");
    if (s->tie_order == VG_TIE_REFERENCE) {
        // vFullScanSortSlots moves distances and rowids together (sqlite-vector.c:2051-2069): carry the rowids through the same exchanges
        VgRefSlots t = s->slots;
        for (int i = 0; i < s->k; ++i) t.pos[(size_t)i] = i;                   // (positions stand in for the slot index)
        const int cnt = t.finish();
        for (int i = 0; i < cnt; ++i) { out_dist[i] = t.dist[(size_t)i]; out_rowids[i] = s->slot_rowid[(size_t)t.pos[(size_t)i]]; }
        *out_count = cnt;
    } else {
        const int cnt = (int)s->best_d.size();
        for (int i = 0; i < cnt; ++i) { out_dist[i] = s->best_d[(size_t)i]; out_rowids[i] = s->best_id[(size_t)i]; }
        *out_count = cnt;
    }
    return VG_OK;
}

extern "C" int vg_slab_scan_all(vg_slab_scan *s, int64_t *out_rows, const float **out_dist, const int64_t **out_rowids) {
    if (!s || !out_rows || !out_dist || !out_rowids) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_all: NULL argument");
    if (!s->finished || s->k != 0) return vg_fail(VG_ERR_INVALID, "vg_slab_scan_all: a finished k = 0 scan is needed");
    *out_rows = (int64_t)s->all_d.size();
    *out_dist = s->all_d.data();
    *out_rowids = s->all_id.data();
    return VG_OK;
}

extern "C" void vg_slab_scan_destroy(vg_slab_scan *s) {
    if (!s) return;
    if (s->busy) { s->worker.join(); s->busy = false; }
    for (int i = 0; i < 2; ++i) if (s->slab[i]) vg_corpus_destroy(s->slab[i]);
    delete s;
}

extern "C" int vg_device_memory(int device, long long *out_free_bytes, long long *out_total_bytes) {
    if (!out_free_bytes || !out_total_bytes) return vg_fail(VG_ERR_INVALID, "vg_device_memory: NULL argument");
    size_t f = 0, t = 0;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemGetInfo(&f, &t));
    *out_free_bytes = (long long)f;
    *out_total_bytes = (long long)t;
    return VG_OK;
}
