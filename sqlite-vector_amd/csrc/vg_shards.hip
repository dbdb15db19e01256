// vg_shards.hip - one logical corpus spread over several devices of ONE process (the SQLite extension's case: a
// connection lives in one process).  Host code only: every distance is computed by the per-device
// corpora of vg_api.hip; this file distributes rows, puts all shards in flight and merges their candidate keys.
//
// Distribution is block-cyclic in scan order: global position g lives in block b = g / B, on shard b % S, at local
// position (b / S) * B + g % B.  Local order is therefore monotone in global order, so each shard's own
// (distance, local position) ranking is consistent with the global (distance, scan position) contract and the
// merge only has to compare (distance image, global position) - the result is bit-identical to one big shard.
// Rows can be appended as they stream out of sqlite3_step() without knowing the final count (a contiguous
// row-range split would need it), and every device gets work as soon as B * S rows exist.
//
// The exchange step - S x 64 candidate keys, 512 B per shard - has two forms, chosen per handle (vg_shards_set_gather, the
// VECTORGPU_SHARD_GATHER environment variable: host | rccl):
//   host  (default) every shard copies its 64 keys to pinned host memory behind its scan, one host thread waits for the S streams;
//   rccl            north_star's "RCCL gather of per-shard candidates over xGMI": one communicator per device of this process
//                   (ncclCommInitAll), ONE grouped ncclAllGather of 64 keys per shard on the scan streams, then a single 512 B x S
//                   copy from the first device.  librccl.so is dlopen'ed on first use (the library has no link-time dependency
//                   on it); the devices of a handle must be distinct (RCCL refuses two ranks on one device) - otherwise, or if
//                   the library / a call fails, the handle stays with the host gather.
// Both produce the same S x 64 keys, so the merge - and the result - is identical bit for bit; which one is faster is a
// measurement (tools/shards_gather_bench.py): the payload is latency-bound either way.  The one-process-per-GPU variant of the
// same exchange (torch.distributed / RCCL) is shard.py.
#include "vg_internal.h"
#include "vg_refslots.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" void vg_set_last_error_(const char *msg);          // vg_api.hip (thread-local error slot)

#define VG_WAVE_KEYS 64

struct vg_shards {
    int S = 0;
    int64_t B = 0;
    int vtype = 0, dim = 0, es = 0;
    int64_t n_rows = 0;
    int64_t rowid_base = 1;
    int tie_order = VG_TIE_POSITION;
    std::vector<vg_corpus *> sh;
    std::vector<int> devices;
    // ---- the RCCL form of the candidate gather (see the file header)
    int gather_mode = 0;                       // 0 = host, 1 = rccl
    std::vector<void *> comms;                 // ncclComm_t per shard (empty until first use)
    std::vector<uint64_t *> d_gather;          // per shard: S x 64 keys (the all-gather's receive buffer)
    uint64_t *h_gather = nullptr;              // pinned: S x 64 keys
    bool rccl_failed = false;
    unsigned long long gather_calls[2] = {0, 0};
    // tie_order = reference (see shards_scan_topk_fused): the same counters and the same "ties were seen recently" rule as one corpus'
    int ref_hot = 0;
    unsigned long long ref_stats[4] = {0, 0, 0, 0};   // reference-order scans | with a tie among the k+1 best | fused replays | store-mode replays
    struct ShardPool *pool = nullptr;          // persistent host threads, one per shard behind the first (see pool_run)
    int threaded = 0;                          // per-query issue + collect of the shards on those threads
};

// ---- librccl.so, resolved at run time
namespace {
typedef int (*nccl_comm_init_all_t)(void **, int, const int *);
typedef int (*nccl_all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_group_t)(void);
typedef int (*nccl_comm_destroy_t)(void *);
typedef const char *(*nccl_error_string_t)(int);
struct RcclApi {
    void *lib = nullptr;
    nccl_comm_init_all_t comm_init_all = nullptr;
    nccl_all_gather_t all_gather = nullptr;
    nccl_group_t group_start = nullptr, group_end = nullptr;
    nccl_comm_destroy_t comm_destroy = nullptr;
    nccl_error_string_t error_string = nullptr;
    bool tried = false;
};
RcclApi g_rccl;
const int kNcclUint64 = 5;                      // ncclUint64 (rccl.h ncclDataType_t)
bool rccl_load() {
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!g_rccl.lib) return false;
    g_rccl.comm_init_all = (nccl_comm_init_all_t)dlsym(g_rccl.lib, "ncclCommInitAll");
    g_rccl.all_gather = (nccl_all_gather_t)dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.group_start = (nccl_group_t)dlsym(g_rccl.lib, "ncclGroupStart");
    g_rccl.group_end = (nccl_group_t)dlsym(g_rccl.lib, "ncclGroupEnd");
    g_rccl.comm_destroy = (nccl_comm_destroy_t)dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.error_string = (nccl_error_string_t)dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.comm_init_all || !g_rccl.all_gather || !g_rccl.group_start || !g_rccl.group_end || !g_rccl.comm_destroy) {
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
    }
    return g_rccl.lib != nullptr;
}
}

static int fail(int code, const char *msg) {
    vg_set_last_error_(msg);
    return code;
}
int vg_scan_topk_enqueue_dev(vg_corpus *c, int metric, const void *query, int k);    // vg_api.hip: keys stay in c->d_keys

static inline void locate(const vg_shards *s, int64_t g, int *shard, int64_t *local) {
    const int64_t b = g / s->B;
    *shard = (int)(b % s->S);
    *local = (b / s->S) * s->B + g % s->B;
}

static inline int64_t global_of(const vg_shards *s, int shard, int64_t local) {
    return ((local / s->B) * s->S + shard) * s->B + local % s->B;
}

// ---- PERSISTENT host threads for the per-query work (one thread issued S x (hipSetDevice + H2D + launches) and then collected
// S times - on eight real devices ~0.1 ms of a 2.4 ms scan of a 12.5M-row shard).  Shard i >= 1 has a thread of its own, bound to the
// shard's device once; a query wakes them all, the calling thread serves shard 0 itself, and every thread enqueues AND collects its
// shard.  On by default when the shards sit on more than one device; VECTORGPU_SHARD_THREADS=1 / 0 forces it (the tests run it over
// logical shards of one device).
struct ShardPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable go, done;
    uint64_t epoch = 0;
    int pending = 0;
    bool quit = false;
    std::function<int(int)> job;
    std::vector<int> rc;
    std::vector<std::string> msg;
};
static void pool_worker(vg_shards *s, ShardPool *p, int shard) {
    hipSetDevice(s->devices[(size_t)shard]);
    uint64_t seen = 0;
    for (;;) {
        std::unique_lock<std::mutex> lk(p->mu);
        p->go.wait(lk, [&] { return p->quit || p->epoch != seen; });
        if (p->quit) return;
        seen = p->epoch;
        lk.unlock();
        const int r = p->job(shard);
        const std::string m = r != VG_OK ? vg_last_error() : "";
        lk.lock();
        p->rc[(size_t)shard] = r;
        if (r != VG_OK) p->msg[(size_t)shard] = m;
        if (--p->pending == 0) p->done.notify_one();
    }
}
static void pool_release(vg_shards *s) {
    ShardPool *p = s->pool;
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->mu); p->quit = true; }
    p->go.notify_all();
    for (auto &t : p->th) t.join();
    delete p;
    s->pool = nullptr;
}
// fn(shard) for every shard at once; the first failure's code and message are re-raised on the calling thread
template <typename F>
static int pool_run(vg_shards *s, F fn) {
    if (s->S == 1) return fn(0);
    if (!s->pool) {
        s->pool = new ShardPool();
        s->pool->rc.assign((size_t)s->S, VG_OK);
        s->pool->msg.assign((size_t)s->S, std::string());
        for (int i = 1; i < s->S; ++i) s->pool->th.emplace_back(pool_worker, s, s->pool, i);
    }
    ShardPool *p = s->pool;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->job = fn;
        p->pending = s->S - 1;
        std::fill(p->rc.begin(), p->rc.end(), VG_OK);
        ++p->epoch;
    }
    p->go.notify_all();
    const int r0 = fn(0);
    const std::string m0 = r0 != VG_OK ? vg_last_error() : "";
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->done.wait(lk, [&] { return p->pending == 0; });
    }
    if (r0 != VG_OK) return fail(r0, m0.c_str());
    for (int i = 1; i < s->S; ++i)
        if (p->rc[(size_t)i] != VG_OK) return fail(p->rc[(size_t)i], p->msg[(size_t)i].c_str());
    return VG_OK;
}

// run fn(shard) for every shard concurrently (one host thread each: the per-corpus calls block on their stream);
// the first failure's code and message are re-raised on the calling thread
template <typename F>
static int for_each_shard(vg_shards *s, F fn) {
    if (s->S == 1) return fn(0);
    std::vector<int> rc((size_t)s->S, VG_OK);
    std::vector<std::string> msg((size_t)s->S);
    std::vector<std::thread> th;
    for (int i = 0; i < s->S; ++i)
        th.emplace_back([&, i] {
            rc[(size_t)i] = fn(i);
            if (rc[(size_t)i] != VG_OK) msg[(size_t)i] = vg_last_error();
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < s->S; ++i)
        if (rc[(size_t)i] != VG_OK) return fail(rc[(size_t)i], msg[(size_t)i].c_str());
    return VG_OK;
}

extern "C" int vg_shards_create(const int *devices, int n_devices, int vtype, int dim, int64_t block_rows, vg_shards **out) {
    if (!out) return fail(VG_ERR_INVALID, "vg_shards_create: out is NULL");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(VG_ERR_INVALID, "vg_shards_create: 1..64 shards");
    vg_shards *s = new vg_shards();
    s->S = n_devices;
    s->B = block_rows > 0 ? block_rows : 65536;
    s->vtype = vtype;
    s->dim = dim;
    {
        vg_switches_read();                                           // (a shard set, like a corpus, takes the environment as it is now)
        s->gather_mode = vg_sw(SW_VECTORGPU_SHARD_GATHER, 0);
    }
    for (int i = 0; i < n_devices; ++i) {
        vg_corpus *c = nullptr;
        s->devices.push_back(devices ? devices[i] : i);
        int rc = vg_corpus_create(devices ? devices[i] : i, vtype, dim, 0, &c);
        if (rc != VG_OK) {
            for (auto *p : s->sh) vg_corpus_destroy(p);
            delete s;
            return rc;
        }
        s->sh.push_back(c);
    }
    s->es = (vtype == VG_TYPE_F32) ? 4 : (vtype == VG_TYPE_F16 || vtype == VG_TYPE_BF16) ? 2 : 1;
    {
        bool many = false;
        for (int d : s->devices) many = many || d != s->devices[0];
        s->threaded = vg_sw(SW_VECTORGPU_SHARD_THREADS, many ? 1 : 0) != 0 && s->S > 1;
    }
    *out = s;
    return VG_OK;
}

static void rccl_release(vg_shards *s) {
    for (size_t i = 0; i < s->comms.size(); ++i) if (s->comms[i] && g_rccl.comm_destroy) g_rccl.comm_destroy(s->comms[i]);
    s->comms.clear();
    for (size_t i = 0; i < s->d_gather.size(); ++i)
        if (s->d_gather[i]) { hipSetDevice(s->devices[i]); hipFree(s->d_gather[i]); }
    s->d_gather.clear();
    if (s->h_gather) { hipHostFree(s->h_gather); s->h_gather = nullptr; }
}

extern "C" void vg_shards_destroy(vg_shards *s) {
    if (!s) return;
    pool_release(s);
    rccl_release(s);
    for (auto *p : s->sh) vg_corpus_destroy(p);
    delete s;
}

// communicators + receive buffers on first use; false (and the handle falls back to the host gather for good) when RCCL cannot
// serve this handle
static bool rccl_ready(vg_shards *s) {
    if (s->rccl_failed) return false;
    if (!s->comms.empty()) return true;
    bool distinct = true;
    for (int i = 0; i < s->S; ++i) for (int j = 0; j < i; ++j) distinct &= s->devices[(size_t)i] != s->devices[(size_t)j];
    if (!distinct || !rccl_load()) { s->rccl_failed = true; return false; }
    s->comms.assign((size_t)s->S, nullptr);
    if (g_rccl.comm_init_all(s->comms.data(), s->S, s->devices.data()) != 0) { s->comms.clear(); s->rccl_failed = true; return false; }
    s->d_gather.assign((size_t)s->S, nullptr);
    bool ok = true;
    for (int i = 0; i < s->S && ok; ++i) {
        ok = hipSetDevice(s->devices[(size_t)i]) == hipSuccess &&
             hipMalloc(&s->d_gather[(size_t)i], (size_t)s->S * VG_WAVE_KEYS * sizeof(uint64_t)) == hipSuccess;
    }
    ok = ok && hipHostMalloc(&s->h_gather, (size_t)s->S * VG_WAVE_KEYS * sizeof(uint64_t)) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); rccl_release(s); s->rccl_failed = true; return false; }
    return true;
}

// The RCCL form of "scan every shard, bring the S x 64 candidate keys to the host": scans enqueued on the shards' streams with the
// keys left on their devices, one grouped all-gather on those streams, one copy from the first device.  VG_OK, or an error after
// which the caller repeats the query through the host gather.
static int rccl_scan_and_gather(vg_shards *s, int metric, const void *query, int kk, uint64_t *out_keys) {
    int rc = VG_OK;
    for (int i = 0; i < s->S && rc == VG_OK; ++i) rc = vg_scan_topk_enqueue_dev(s->sh[(size_t)i], metric, query, kk);
    if (rc != VG_OK) return rc;
    int nrc = g_rccl.group_start();
    for (int i = 0; i < s->S && nrc == 0; ++i) {
        vg_corpus *c = s->sh[(size_t)i];
        nrc = g_rccl.all_gather(c->d_keys, s->d_gather[(size_t)i], VG_WAVE_KEYS, kNcclUint64, s->comms[(size_t)i], c->stream);
    }
    const int erc = g_rccl.group_end();
    if (nrc != 0 || erc != 0) {
        for (int i = 0; i < s->S; ++i) { hipSetDevice(s->devices[(size_t)i]); hipStreamSynchronize(s->sh[(size_t)i]->stream); s->sh[(size_t)i]->enqueued = false; }
        return fail(VG_ERR_HIP, (g_rccl.error_string ? g_rccl.error_string(nrc ? nrc : erc) : "RCCL all-gather failed"));
    }
    vg_corpus *c0 = s->sh[0];
    HIP_TRY(hipSetDevice(s->devices[0]));
    HIP_TRY(hipMemcpyAsync(s->h_gather, s->d_gather[0], (size_t)s->S * VG_WAVE_KEYS * sizeof(uint64_t), hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipStreamSynchronize(c0->stream));                 // (every shard's keys arrived on device 0: all scans are done)
    for (int i = 0; i < s->S; ++i) {
        vg_corpus *c = s->sh[(size_t)i];
        c->enqueued = false;
        if (c->profiling && i > 0) {                           // its own all-gather kernel (and the events behind it) may still be running
            HIP_TRY(hipSetDevice(s->devices[(size_t)i]));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        vg_collect_timing(c);
    }
    memcpy(out_keys, s->h_gather, (size_t)s->S * VG_WAVE_KEYS * sizeof(uint64_t));
    ++s->gather_calls[1];
    return VG_OK;
}

extern "C" int vg_shards_set_gather(vg_shards *s, int mode) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    if (mode != 0 && mode != 1) return fail(VG_ERR_INVALID, "vg_shards_set_gather: 0 = host, 1 = rccl");
    s->gather_mode = mode;
    if (mode == 1 && s->rccl_failed) {                         // asked again explicitly: try again, with communicators of its own
        rccl_release(s);                                       // (the ones of a failed group call may be broken)
        s->rccl_failed = false;
    }
    return VG_OK;
}
// [0] queries answered through the host gather, [1] through the RCCL all-gather; returns the mode in effect (1 only when RCCL serves)
extern "C" int vg_shards_gather_stats(vg_shards *s, unsigned long long *out2) {
    if (!s) return 0;
    if (out2) { out2[0] = s->gather_calls[0]; out2[1] = s->gather_calls[1]; }
    return (s->gather_mode == 1 && !s->rccl_failed) ? 1 : 0;
}

extern "C" int vg_shards_clear(vg_shards *s) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    for (auto *p : s->sh) {
        int rc = vg_corpus_clear(p);
        if (rc != VG_OK) return rc;
    }
    s->n_rows = 0;
    return VG_OK;
}

extern "C" int vg_shards_count(const vg_shards *s) { return s ? s->S : 0; }
extern "C" int64_t vg_shards_rows(const vg_shards *s) { return s ? s->n_rows : 0; }
extern "C" vg_corpus *vg_shards_shard(const vg_shards *s, int i) { return (s && i >= 0 && i < s->S) ? s->sh[(size_t)i] : nullptr; }

extern "C" int vg_shards_reserve(vg_shards *s, int64_t total_rows) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    if (total_rows <= 0) return VG_OK;
    const int64_t blocks = (total_rows + s->B - 1) / s->B;
    for (int i = 0; i < s->S; ++i) {
        const int64_t mine = (blocks + s->S - 1 - i) / s->S;                   // blocks i, i+S, i+2S, ...
        if (mine <= 0) continue;
        const int64_t rows = std::min<int64_t>(mine * s->B, total_rows);
        int rc = vg_corpus_reserve(s->sh[(size_t)i], rows);
        if (rc != VG_OK) return rc;
    }
    return VG_OK;
}

extern "C" int vg_corpus_clone(const vg_corpus *src, vg_corpus **out);
extern "C" int vg_shards_clone(const vg_shards *src, vg_shards **out) {
    if (!src || !out) return fail(VG_ERR_INVALID, "vg_shards_clone: NULL argument");
    *out = nullptr;
    vg_shards *s = nullptr;
    int rc = vg_shards_create(src->devices.data(), src->S, src->vtype, src->dim, src->B, &s);
    if (rc != VG_OK) return rc;
    for (int i = 0; i < src->S; ++i) {
        vg_corpus *c = nullptr;
        rc = vg_corpus_clone(src->sh[(size_t)i], &c);
        if (rc != VG_OK) { vg_shards_destroy(s); return rc; }
        vg_corpus_destroy(s->sh[(size_t)i]);                                  // (the empty corpus vg_shards_create made)
        s->sh[(size_t)i] = c;
    }
    s->n_rows = src->n_rows;
    s->rowid_base = src->rowid_base;
    s->tie_order = src->tie_order;
    s->gather_mode = src->gather_mode;
    *out = s;
    return VG_OK;
}

extern "C" int vg_corpus_trim(vg_corpus *c);
extern "C" int vg_shards_trim(vg_shards *s) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    for (int i = 0; i < s->S; ++i) {
        int rc = vg_corpus_trim(s->sh[(size_t)i]);
        if (rc != VG_OK) return rc;
    }
    return VG_OK;
}

extern "C" int vg_shards_set_rowid_base(vg_shards *s, int64_t base) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    s->rowid_base = base;
    if (s->S == 1) return vg_corpus_set_rowid_base(s->sh[0], base);
    return VG_OK;
}

// rows (or [rowid | vector] records when record_mode) are dealt out block by block
static int append_common(vg_shards *s, const void *host, int64_t n, int64_t stride, const int64_t *rowids, bool record_mode) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    if (n == 0) return VG_OK;
    if (!host || n < 0) return fail(VG_ERR_INVALID, "vg_shards_append: bad rows pointer / count");
    const uint8_t *p = (const uint8_t *)host;
    std::vector<int64_t> gen;
    int64_t done = 0;
    while (done < n) {
        const int64_t g = s->n_rows;
        int shard;
        int64_t local;
        locate(s, g, &shard, &local);
        const int64_t take = std::min<int64_t>(s->B - g % s->B, n - done);
        int rc;
        if (record_mode) {
            rc = vg_corpus_append_records(s->sh[(size_t)shard], p + done * stride, take);
        } else {
            const int64_t *ids = rowids ? rowids + done : nullptr;
            if (!ids && s->S > 1) {                    // implicit rowids follow the GLOBAL position, not the shard's
                gen.resize((size_t)take);
                for (int64_t i = 0; i < take; ++i) gen[(size_t)i] = s->rowid_base + g + i;
                ids = gen.data();
            }
            rc = vg_corpus_append(s->sh[(size_t)shard], p + done * stride, take, stride, ids);
        }
        if (rc != VG_OK) return rc;
        s->n_rows += take;
        done += take;
    }
    return VG_OK;
}

extern "C" int vg_shards_append(vg_shards *s, const void *host_rows, int64_t n_rows, int64_t row_stride_bytes, const int64_t *rowids) {
    return append_common(s, host_rows, n_rows, row_stride_bytes, rowids, false);
}

extern "C" int vg_shards_append_records(vg_shards *s, const void *host_records, int64_t n_records) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    return append_common(s, host_records, n_records, 8 + (int64_t)s->dim, nullptr, true);
}

extern "C" int64_t vg_shards_rowid_at(const vg_shards *s, int64_t position) {
    if (!s || position < 0 || position >= s->n_rows) return 0;
    int shard;
    int64_t local;
    locate(s, position, &shard, &local);
    return vg_corpus_rowid_at(s->sh[(size_t)shard], local);
}

extern "C" int vg_shards_rowids(const vg_shards *s, int64_t pos0, int64_t n, int64_t *out) {
    if (!s || !out || pos0 < 0 || n < 0 || pos0 + n > s->n_rows) return fail(VG_ERR_INVALID, "vg_shards_rowids: bad range");
    int64_t g = pos0;
    while (g < pos0 + n) {                                   // block by block: one shard, consecutive local positions
        int shard;
        int64_t local;
        locate(s, g, &shard, &local);
        const int64_t take = std::min<int64_t>(s->B - g % s->B, pos0 + n - g);
        const vg_corpus *c = s->sh[(size_t)shard];
        for (int64_t i = 0; i < take; ++i) out[g - pos0 + i] = vg_corpus_rowid_at(c, local + i);
        g += take;
    }
    return VG_OK;
}

// Row maintenance (vg_corpus_patch_rows / _delete_rows / _find_rowid).  A deletion shifts every later row by one position, i.e.
// across the block-cyclic deal of several shards: only single-shard handles are served, the caller re-stages otherwise.
extern "C" int64_t vg_shards_find_rowid(const vg_shards *s, int64_t rowid) {
    if (!s) return -1;
    if (s->S != 1) return -2;
    return vg_corpus_find_rowid(s->sh[0], rowid);
}
extern "C" int vg_shards_patch_rows(vg_shards *s, const int64_t *positions, int64_t n, const void *host_rows, int64_t row_stride_bytes) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    if (s->S != 1) return fail(VG_ERR_UNSUPPORTED, "row maintenance needs a single-shard corpus");
    return vg_corpus_patch_rows(s->sh[0], positions, n, host_rows, row_stride_bytes);
}
extern "C" int vg_shards_delete_rows(vg_shards *s, const int64_t *positions, int64_t n) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    if (s->S != 1) return fail(VG_ERR_UNSUPPORTED, "row maintenance needs a single-shard corpus");
    int rc = vg_corpus_delete_rows(s->sh[0], positions, n);
    if (rc == VG_OK) s->n_rows = vg_corpus_rows(s->sh[0]);
    return rc;
}

extern "C" int vg_shards_set_scan_filter(vg_shards *s, int mode) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    for (auto *p : s->sh) {
        int rc = vg_corpus_set_scan_filter(p, mode);
        if (rc != VG_OK) return rc;
    }
    return VG_OK;
}

struct Cand { uint32_t img; int64_t gpos; int shard; uint32_t local; };

static inline bool cand_less(const Cand &a, const Cand &b) { return a.img != b.img ? a.img < b.img : a.gpos < b.gpos; }

// merge per-shard ascending key lists (list i = shard i, `len` keys each, `counts[i]` valid) into the global top-k.
// ref_k >= 0: the lists were taken with one more slot than asked for (tie_order = reference, see vg_scan_topk_reference): when the
// best ref_k + 1 distances are pairwise distinct the reference's own result IS the first ref_k in ascending order and that is what
// comes back; when they hold a tie nothing is written and -1 comes back (the caller replays the reference's slots).
static int merge_lists(const vg_shards *s, const uint64_t *keys, int len, const int *counts, int k, int64_t *out_rowids,
                       double *out_dist, int ref_k = -1) {
    std::vector<Cand> all;
    for (int i = 0; i < s->S; ++i)
        for (int j = 0; j < counts[i]; ++j) {
            const uint64_t key = keys[(size_t)i * len + j];
            if (key == VG_KEY_EMPTY) break;
            const uint32_t local = vg_key_position(key);
            all.push_back(Cand{(uint32_t)(key >> 32), global_of(s, i, (int64_t)local), i, local});
        }
    size_t take = std::min<size_t>((size_t)k, all.size());
    std::partial_sort(all.begin(), all.begin() + (long)take, all.end(), cand_less);
    if (ref_k >= 0) {
        for (size_t i = 1; i < take; ++i) if (all[i].img == all[i - 1].img) return -1;
        take = std::min<size_t>(take, (size_t)ref_k);
    }
    for (size_t i = 0; i < take; ++i) {
        out_dist[i] = (double)vg_key_distance((uint64_t)all[i].img << 32);
        out_rowids[i] = vg_corpus_rowid_at(s->sh[(size_t)all[i].shard], (int64_t)all[i].local);
    }
    return (int)take;
}

// ---- tie_order = reference over several shards: the same replay (vg_refslots.h), its stream assembled in GLOBAL scan
// order from the shards' resident distances
extern "C" int vg_shards_set_tie_order(vg_shards *s, int mode) {
    if (!s) return fail(VG_ERR_INVALID, "shards handle is NULL");
    for (auto *p : s->sh) {
        int rc = vg_corpus_set_tie_order(p, mode);
        if (rc != VG_OK) return rc;
    }
    s->tie_order = mode;
    return VG_OK;
}

namespace {
struct ShardsSrc {
    vg_shards *s;
    std::vector<uint64_t> pairs;
    static const int64_t kCap = 1 << 17;
    int fetch(int64_t g0, int64_t cnt, float *out) {
        int64_t g = g0;
        while (g < g0 + cnt) {
            int shard;
            int64_t local;
            locate(s, g, &shard, &local);
            const int64_t take = std::min<int64_t>(s->B - g % s->B, g0 + cnt - g);
            int rc = vg_resident_distances_fetch(s->sh[(size_t)shard], local, take, out + (g - g0));
            if (rc != VG_OK) return rc;
            g += take;
        }
        return VG_OK;
    }
    int below(int64_t g0, float bound, std::vector<VgRefCand> &out, bool *overflow) {
        pairs.resize((size_t)kCap);
        const int64_t pb = g0 / s->B, rem = g0 % s->B;
        for (int i = 0; i < s->S; ++i) {
            // local rows of shard i in front of global position g0: its whole blocks below block pb, + rem when pb is its own
            const int64_t full = (pb + s->S - 1 - i) / s->S;
            const int64_t local_from = full * s->B + ((int)(pb % s->S) == i ? rem : 0);
            int64_t count = 0;
            int rc = vg_resident_distances_below(s->sh[(size_t)i], local_from, bound, pairs.data(), kCap, &count);
            if (rc != VG_OK) return rc;
            if (count > kCap) { *overflow = true; return VG_OK; }
            for (int64_t j = 0; j < count; ++j) {
                const uint32_t bits = (uint32_t)pairs[(size_t)j];
                float d;
                memcpy(&d, &bits, 4);
                out.push_back(VgRefCand{global_of(s, i, (int64_t)(pairs[(size_t)j] >> 32)), d});
            }
        }
        return VG_OK;
    }
};
}

// the store-mode replay: every shard scans again leaving all its distances resident, the host replays a prefix and asks the shards
// for the later rows below the bound (vg_reforder.hip).  What is left for k > 64, rows too long for an emitting kernel and candidate
// streams that overflow.
static int shards_scan_topk_reference(vg_shards *s, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist,
                                      int *out_count) {
    int rc = VG_OK;
    for (int i = 0; i < s->S && rc == VG_OK; ++i) rc = vg_scan_distances_resident(s->sh[(size_t)i], metric, query);   // all shards in flight
    if (rc != VG_OK) return rc;
    ShardsSrc src{s, {}};
    VgRefSlots slots;
    if ((rc = vg_ref_replay(src, s->n_rows, k, slots)) != VG_OK) return rc;
    const int cnt = slots.finish();
    for (int i = 0; i < cnt; ++i) {
        out_dist[i] = slots.dist[(size_t)i];
        out_rowids[i] = vg_shards_rowid_at(s, slots.pos[(size_t)i]);
    }
    *out_count = cnt;
    ++s->ref_stats[3];
    return VG_OK;
}

// The reference's slots replayed over what the shards' emitting launches left behind (vg_scan_topk_reference does the same for one
// corpus).  Every shard ran its own prefix pass over ITS first P_i rows and emitted, behind them, every row that beats the k-th best
// of the shard's EARLIER rows: local order is monotone in global order, so those rows precede the row globally too and the union of
// the shards' streams is a superset of the rows that can enter the slots.  Offering a row that cannot enter is a no-op, so replaying
// the union in GLOBAL scan order ends in exactly the reference's slots - no second scan, whatever the number of shards.
// *overflow: a shard's stream did not fit its buffer / a shard could not emit (the caller takes the store-mode replay).
static int shards_replay_emitted(vg_shards *s, int k, VgRefSlots &slots, bool *overflow) {
    struct Part { const float *prefix = nullptr; int64_t P = 0; unsigned long long *pairs = nullptr; unsigned long long count = 0; };
    std::vector<Part> parts((size_t)s->S);
    *overflow = false;
    for (int i = 0; i < s->S; ++i) {
        vg_corpus *c = s->sh[(size_t)i];
        if (c->n_rows == 0) continue;
        if (c->ref_prefix_rows <= 0) { *overflow = true; return VG_OK; }
        int rc = vg_ref_emitted_enqueue(c);                    // every shard's copies in flight before the first wait
        if (rc != VG_OK) return rc;
    }
    struct GCand { int64_t gpos; float d; };
    std::vector<GCand> cand;
    int64_t max_p = 0;
    for (int i = 0; i < s->S; ++i) {
        vg_corpus *c = s->sh[(size_t)i];
        if (c->n_rows == 0) continue;
        Part &pt = parts[(size_t)i];
        bool ovf = false;
        int rc = vg_ref_emitted_wait(c, &pt.prefix, &pt.P, &pt.pairs, &pt.count, &ovf);
        if (rc != VG_OK) return rc;
        if (ovf) *overflow = true;                             // (keep draining the other shards' streams: their copies are in flight)
        max_p = std::max(max_p, pt.P);
    }
    if (*overflow) return VG_OK;
    for (int i = 0; i < s->S; ++i) {
        const Part &pt = parts[(size_t)i];
        for (unsigned long long j = 0; j < pt.count; ++j) {
            const int64_t local = (int64_t)(pt.pairs[j] >> 32);
            if (local < pt.P) continue;                        // (the main pass covers the prefix rows again)
            const uint32_t bits = (uint32_t)pt.pairs[j];
            float d;
            memcpy(&d, &bits, 4);
            cand.push_back(GCand{global_of(s, i, local), d});
        }
    }
    std::sort(cand.begin(), cand.end(), [](const GCand &x, const GCand &y) { return x.gpos < y.gpos; });
    slots.init(k);
    // the prefixes, block by block in global order (local block lb of shard i is global block lb * S + i), the candidates in between
    size_t ci = 0;
    for (int64_t lb = 0; lb * s->B < max_p; ++lb)
        for (int i = 0; i < s->S; ++i) {
            const Part &pt = parts[(size_t)i];
            const int64_t l0 = lb * s->B;
            if (l0 >= pt.P) continue;
            const int64_t len = std::min<int64_t>(s->B, pt.P - l0);
            const int64_t g0 = global_of(s, i, l0);
            for (; ci < cand.size() && cand[ci].gpos < g0; ++ci) slots.offer(cand[ci].d, cand[ci].gpos);
            vg_ref_offer_run(slots, pt.prefix + l0, len, g0);
        }
    for (; ci < cand.size(); ++ci) slots.offer(cand[ci].d, cand[ci].gpos);
    return VG_OK;
}

// tie_order = reference for k <= 64 over several shards, the fused form: every shard's ordinary top-k scan with one more list slot;
// a tie among the merged k + 1 best -> the replay above.  The policy is one corpus' (vg_scan_topk_reference): scans through a filter
// kernel always emit, plain-kernel scans only while ties are around (a first tie costs one more scan of every shard, emitting);
// k = 64 has no 65th slot: it runs emitting with k slots and always replays.
static int shards_scan_topk_fused(vg_shards *s, int metric, const void *query, int k, bool force_emit, int64_t *out_rowids,
                                  double *out_dist, int *out_count) {
    ++s->ref_stats[0];
    const bool always = (k == VG_WAVE_KEYS);
    const int kk = always ? k : k + 1;
    vg_corpus *probe = nullptr;
    for (auto *c : s->sh) if (c->n_rows > 0) { probe = c; break; }
    bool emit = force_emit || always || s->ref_hot > 0 || (probe && vg_scan_filter_would_serve(probe, metric, kk));
    std::vector<uint64_t> keys((size_t)s->S * VG_WAVE_KEYS);
    std::vector<int> counts((size_t)s->S, VG_WAVE_KEYS);
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc = -1;
        if (!emit && s->gather_mode == 1 && rccl_ready(s)) {      // (an emitting scan leaves more than 64 keys behind: host gather)
            rc = rccl_scan_and_gather(s, metric, query, kk, keys.data());
            if (rc != VG_OK) s->rccl_failed = true;
        }
        if (rc != VG_OK) {
            rc = VG_OK;
            if (s->threaded) {
                rc = pool_run(s, [&](int i) {
                    const int r = vg_scan_topk_enqueue_plan(s->sh[(size_t)i], metric, query, kk, emit);
                    return r != VG_OK ? r : vg_scan_topk_collect(s->sh[(size_t)i], &keys[(size_t)i * VG_WAVE_KEYS]);
                });
            } else {
                for (int i = 0; i < s->S && rc == VG_OK; ++i) rc = vg_scan_topk_enqueue_plan(s->sh[(size_t)i], metric, query, kk, emit);
                for (int i = 0; i < s->S; ++i) {
                    int rc2 = vg_scan_topk_collect(s->sh[(size_t)i], &keys[(size_t)i * VG_WAVE_KEYS]);
                    if (rc == VG_OK) rc = rc2;
                }
            }
            if (rc != VG_OK) return rc;
            ++s->gather_calls[0];
        }
        if (!always) {
            const int got = merge_lists(s, keys.data(), VG_WAVE_KEYS, counts.data(), kk, out_rowids, out_dist, k);
            if (got >= 0) {
                *out_count = got;
                if (s->ref_hot > 0) --s->ref_hot;
                return VG_OK;
            }
            s->ref_hot = 16;
        }
        if (attempt == 0) ++s->ref_stats[1];
        if (emit) {
            VgRefSlots slots;
            bool overflow = false;
            rc = shards_replay_emitted(s, k, slots, &overflow);
            if (rc != VG_OK) return rc;
            if (overflow) break;
            const int n = slots.finish();
            for (int i = 0; i < n; ++i) {
                out_dist[i] = slots.dist[(size_t)i];
                out_rowids[i] = vg_shards_rowid_at(s, slots.pos[(size_t)i]);
            }
            *out_count = n;
            ++s->ref_stats[2];
            return VG_OK;
        }
        emit = true;                                               // a tie and nothing emitted: scan again, emitting
    }
    return shards_scan_topk_reference(s, metric, query, k, out_rowids, out_dist, out_count);
}

extern "C" int vg_corpus_device_bytes(const vg_corpus *c, long long *out3);
extern "C" int vg_shards_device_bytes(const vg_shards *s, long long *out3) {
    if (!s || !out3) return fail(VG_ERR_INVALID, "vg_shards_device_bytes: NULL argument");
    out3[0] = out3[1] = out3[2] = 0;
    for (int i = 0; i < s->S; ++i) {
        long long one[3] = {0, 0, 0};
        const int rc = vg_corpus_device_bytes(s->sh[(size_t)i], one);
        if (rc != VG_OK) return rc;
        for (int j = 0; j < 3; ++j) out3[j] += one[j];
    }
    return VG_OK;
}

extern "C" int vg_shards_threaded(const vg_shards *s) { return s ? s->threaded : 0; }

extern "C" int vg_shards_tie_stats(const vg_shards *s, unsigned long long *out4) {
    if (!s || !out4) return fail(VG_ERR_INVALID, "vg_shards_tie_stats: NULL argument");
    if (s->S == 1) return vg_corpus_tie_stats(s->sh[0], out4);
    for (int i = 0; i < 4; ++i) out4[i] = s->ref_stats[i];
    return VG_OK;
}

extern "C" int vg_shards_scan_topk(vg_shards *s, int metric, const void *query, int k, int64_t *out_rowids, double *out_dist,
                                   int *out_count) {
    if (!s || !query || !out_count) return fail(VG_ERR_INVALID, "vg_shards_scan_topk: NULL argument");
    *out_count = 0;
    if (s->S == 1 && s->gather_mode == 0) return vg_scan_topk(s->sh[0], metric, query, k, out_rowids, out_dist, out_count);
    if (k <= 0 || s->n_rows == 0) return VG_OK;
    if (!out_rowids || !out_dist) return fail(VG_ERR_INVALID, "vg_shards_scan_topk: NULL output");
    const bool ref = s->tie_order == VG_TIE_REFERENCE;
    if (ref && k > VG_WAVE_KEYS) { ++s->ref_stats[0]; return shards_scan_topk_reference(s, metric, query, k, out_rowids, out_dist, out_count); }
    if (ref) return shards_scan_topk_fused(s, metric, query, k, false, out_rowids, out_dist, out_count);
    if (k <= VG_WAVE_KEYS) {
        // every shard in flight before the first wait: S scans run concurrently, one host thread
        std::vector<uint64_t> keys((size_t)s->S * VG_WAVE_KEYS);
        std::vector<int> counts((size_t)s->S, VG_WAVE_KEYS);
        int rc = -1;
        if (s->gather_mode == 1 && rccl_ready(s)) {
            rc = rccl_scan_and_gather(s, metric, query, k, keys.data());
            if (rc != VG_OK) s->rccl_failed = true;            // (the host gather serves this query and the ones after it)
        }
        if (rc != VG_OK) {
            rc = VG_OK;
            if (s->threaded) {
                rc = pool_run(s, [&](int i) {
                    const int r = vg_scan_topk_enqueue(s->sh[(size_t)i], metric, query, k);
                    return r != VG_OK ? r : vg_scan_topk_collect(s->sh[(size_t)i], &keys[(size_t)i * VG_WAVE_KEYS]);
                });
            } else {
                for (int i = 0; i < s->S && rc == VG_OK; ++i) rc = vg_scan_topk_enqueue(s->sh[(size_t)i], metric, query, k);
                for (int i = 0; i < s->S; ++i) {
                    int rc2 = vg_scan_topk_collect(s->sh[(size_t)i], &keys[(size_t)i * VG_WAVE_KEYS]);
                    if (rc == VG_OK) rc = rc2;
                }
            }
            if (rc != VG_OK) return rc;
            ++s->gather_calls[0];
        }
        *out_count = merge_lists(s, keys.data(), VG_WAVE_KEYS, counts.data(), k, out_rowids, out_dist);
        return VG_OK;
    }
    const int kk = (int)std::min<int64_t>((int64_t)k, s->n_rows);
    std::vector<uint64_t> keys((size_t)s->S * kk);
    std::vector<int> counts((size_t)s->S, 0);
    int rc = for_each_shard(s, [&](int i) {
        return vg_scan_topk_keys(s->sh[(size_t)i], metric, query, kk, &keys[(size_t)i * kk], &counts[(size_t)i]);
    });
    if (rc != VG_OK) return rc;
    *out_count = merge_lists(s, keys.data(), kk, counts.data(), kk, out_rowids, out_dist);
    return VG_OK;
}

extern "C" int vg_shards_scan_topk_batch(vg_shards *s, int metric, const void *queries, int nq, int k, int64_t *out_rowids,
                                         double *out_dist, int *out_counts) {
    if (!s || !queries || !out_counts) return fail(VG_ERR_INVALID, "vg_shards_scan_topk_batch: NULL argument");
    if (s->S == 1) return vg_scan_topk_batch(s->sh[0], metric, queries, nq, k, out_rowids, out_dist, out_counts);
    if (nq <= 0) return VG_OK;
    for (int i = 0; i < nq; ++i) out_counts[i] = 0;
    if (k <= 0 || s->n_rows == 0) return VG_OK;
    if (!out_rowids || !out_dist) return fail(VG_ERR_INVALID, "vg_shards_scan_topk_batch: NULL output");
    const bool ref = s->tie_order == VG_TIE_REFERENCE;      // (one more list slot; only the queries with a tie are answered again, one by one)
    const size_t qbytes = (size_t)s->dim * s->es;
    if (ref && k >= VG_WAVE_KEYS) {                          // no slot to look for a tie with: every query through the single-query form
        for (int q = 0; q < nq; ++q) {
            int rc1 = vg_shards_scan_topk(s, metric, (const uint8_t *)queries + (size_t)q * qbytes, k, out_rowids + (size_t)q * k,
                                          out_dist + (size_t)q * k, &out_counts[q]);
            if (rc1 != VG_OK) return rc1;
        }
        return VG_OK;
    }
    // f32 rows on the f32 matrix cores sum in another order than the single scans the reference order is defined on: a tie under the
    // single scan's arithmetic could go unnoticed in the batch's keys - such batches are answered query by query
    if (ref && s->vtype == VG_TYPE_F32 && !vg_batch_keys_are_scan_exact(s->sh[0], metric, k + 1)) {
        for (int q = 0; q < nq; ++q) {
            int rc1 = shards_scan_topk_fused(s, metric, (const uint8_t *)queries + (size_t)q * qbytes, k, false, out_rowids + (size_t)q * k,
                                             out_dist + (size_t)q * k, &out_counts[q]);
            if (rc1 != VG_OK) return rc1;
        }
        return VG_OK;
    }
    const int kk = (int)std::min<int64_t>((int64_t)k + (ref ? 1 : 0), s->n_rows);
    std::vector<uint64_t> keys((size_t)s->S * nq * kk);
    std::vector<int> counts((size_t)s->S * nq, 0);
    int rc = for_each_shard(s, [&](int i) {
        return vg_scan_topk_batch_keys(s->sh[(size_t)i], metric, queries, nq, kk, &keys[(size_t)i * nq * kk], &counts[(size_t)i * nq]);
    });
    if (rc != VG_OK) return rc;
    std::vector<uint64_t> qkeys((size_t)s->S * kk);
    std::vector<int> qcounts((size_t)s->S);
    for (int q = 0; q < nq; ++q) {
        for (int i = 0; i < s->S; ++i) {
            memcpy(&qkeys[(size_t)i * kk], &keys[((size_t)i * nq + q) * kk], (size_t)kk * sizeof(uint64_t));
            qcounts[(size_t)i] = counts[(size_t)i * nq + q];
        }
        int got = merge_lists(s, qkeys.data(), kk, qcounts.data(), kk, out_rowids + (size_t)q * k, out_dist + (size_t)q * k, ref ? k : -1);
        if (got < 0) {                                       // a tie: this query again through the emitting scans + the replay
            int rc1 = shards_scan_topk_fused(s, metric, (const uint8_t *)queries + (size_t)q * qbytes, k, true, out_rowids + (size_t)q * k,
                                             out_dist + (size_t)q * k, &got);
            if (rc1 != VG_OK) return rc1;
        }
        out_counts[q] = got;
    }
    return VG_OK;
}

extern "C" int vg_shards_scan_distances(vg_shards *s, int metric, const void *query, float *out_dist_host) {
    if (!s || !query || !out_dist_host) return fail(VG_ERR_INVALID, "vg_shards_scan_distances: NULL argument");
    if (s->S == 1) return vg_scan_distances(s->sh[0], metric, query, out_dist_host);
    if (s->n_rows == 0) return VG_OK;
    return for_each_shard(s, [&](int i) {
        vg_corpus *c = s->sh[(size_t)i];
        const int64_t n = vg_corpus_rows(c);
        if (n == 0) return (int)VG_OK;
        std::vector<float> tmp((size_t)n);
        int rc = vg_scan_distances(c, metric, query, tmp.data());
        if (rc != VG_OK) return rc;
        for (int64_t l0 = 0; l0 < n; l0 += s->B) {                                   // local block -> its global place
            const int64_t len = std::min<int64_t>(s->B, n - l0);
            memcpy(out_dist_host + global_of(s, i, l0), &tmp[(size_t)l0], (size_t)len * sizeof(float));
        }
        return (int)VG_OK;
    });
}

extern "C" int vg_shards_minmax(vg_shards *s, float *out_min, float *out_max, int *out_any_negative) {
    if (!s || !out_min || !out_max || !out_any_negative) return fail(VG_ERR_INVALID, "vg_shards_minmax: NULL argument");
    if (s->S == 1) return vg_corpus_minmax(s->sh[0], out_min, out_max, out_any_negative);
    std::vector<float> lo((size_t)s->S), hi((size_t)s->S);
    std::vector<int> neg((size_t)s->S);
    int rc = for_each_shard(s, [&](int i) { return vg_corpus_minmax(s->sh[(size_t)i], &lo[(size_t)i], &hi[(size_t)i], &neg[(size_t)i]); });
    if (rc != VG_OK) return rc;
    *out_min = lo[0]; *out_max = hi[0]; *out_any_negative = neg[0];
    for (int i = 1; i < s->S; ++i) {
        if (lo[(size_t)i] < *out_min) *out_min = lo[(size_t)i];
        if (hi[(size_t)i] > *out_max) *out_max = hi[(size_t)i];
        *out_any_negative |= neg[(size_t)i];
    }
    return VG_OK;
}

extern "C" int vg_shards_quantize_rows(vg_shards *s, float scale, float offset, int qtype, int64_t row0, int64_t n_rows,
                                       uint8_t *out_host) {
    if (!s || !out_host) return fail(VG_ERR_INVALID, "vg_shards_quantize_rows: NULL argument");
    if (s->S == 1) return vg_corpus_quantize_rows(s->sh[0], scale, offset, qtype, row0, n_rows, out_host);
    if (row0 < 0 || n_rows < 0 || row0 + n_rows > s->n_rows) return fail(VG_ERR_INVALID, "vg_shards_quantize_rows: row range out of bounds");
    int64_t g = row0;
    while (g < row0 + n_rows) {
        int shard;
        int64_t local;
        locate(s, g, &shard, &local);
        const int64_t take = std::min<int64_t>(s->B - g % s->B, row0 + n_rows - g);
        int rc = vg_corpus_quantize_rows(s->sh[(size_t)shard], scale, offset, qtype, local, take, out_host + (g - row0) * s->dim);
        if (rc != VG_OK) return rc;
        g += take;
    }
    return VG_OK;
}
