/*
 * vector_ext.c - SQLite loadable extension host for the MI355X scan engine (plain C).
 *
 * Drop-in for sqlite-vector's SQL surface on its hot path: the same entry point (sqlite3_vector_init), the same
 * 13 scalar function names / arities and the same 4 table-valued functions with the same hidden-column schema
 * (reference: sqlite-vector.c:2555-2638, :1830, :98-103), the same persisted formats (_sqliteai_vector keys,
 * vector0_<tbl>_<col> records), the same error texts where a caller could match on them.  What differs is
 * behind the seam the reference calls vcursor_run_callback (sqlite-vector.c:183): rows are staged ONCE into HBM
 * (vg_corpus_*) and every scan is a HIP kernel launch through the C-ABI in include/vectorgpu.h.  No distance is
 * computed in this file; if libvectorgpu.so or a GPU is missing the scan functions raise an SQL error.
 *
 * The C-ABI library is dlopen()ed from the directory this extension was loaded from (or $VECTORGPU_LIB), so the
 * extension itself links only against libc/libdl and talks to SQLite through sqlite3_api_routines.
 *
 * Deliberate deviations from the reference (documented in DESIGN.md):
 *   - result order among EQUAL distances: the reference's own slot-history dependent result, rowid for rowid, for integer
 *     element types (quantized scans, INT8 / UINT8 columns) by default; (distance, scan position) for float columns
 *     (vector_init option tie_order=reference|position / VECTORGPU_TIE_ORDER, see tie_order_for());
 *   - the *_stream modules emit exactly the N rows (the reference emits a spurious leading (0, 0.0) row because
 *     xFilter never advances, sqlite-vector.c:1790-1792);
 *   - the JSON query vector is freed (the reference leaks it, :1771).
 */
#define _GNU_SOURCE            /* dladdr, strcasestr */
#include "sqlite3ext.h"
SQLITE_EXTENSION_INIT1

#include <ctype.h>
#include <dlfcn.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "vectorgpu.h"

#define VECTOR_EXT_VERSION "0.9.23-mi355x.1"     /* tracks the reference release it is a drop-in for (0.9.23) */
#define MAX_TABLES 128
#define DEFAULT_MAX_MEMORY (30 * 1024 * 1024)
#define SQL_BUF 2048
#define STAGE_ROWS 8192                           /* rows per host staging block while filling HBM */

/* hidden/visible column indices of the TVF schema (sqlite-vector.c:98-103) */
enum { COL_TBL = 0, COL_VECTOR = 1, COL_K = 2, COL_MEMIDX = 3, COL_ID = 4, COL_DISTANCE = 5, COL_QUERY = 6 /* batch TVFs only */ };

/* ------------------------------------------------------------------------------------------------ GPU library */

typedef struct {
    void *handle;
    int ready;                  /* every symbol below is bound */
    int (*device_count)(void);
    const char *(*backend_name)(void);
    const char *(*last_error)(void);
    int (*corpus_create)(const int *, int, int, int, int64_t, vg_shards **);
    void (*corpus_destroy)(vg_shards *);
    int (*corpus_clear)(vg_shards *);
    int (*corpus_reserve)(vg_shards *, int64_t);
    int64_t (*corpus_rows)(const vg_shards *);
    int (*corpus_append)(vg_shards *, const void *, int64_t, int64_t, const int64_t *);
    int (*corpus_append_records)(vg_shards *, const void *, int64_t);
    int (*scan_topk)(vg_shards *, int, const void *, int, int64_t *, double *, int *);
    int (*scan_distances)(vg_shards *, int, const void *, float *);
    int (*scan_topk_batch)(vg_shards *, int, const void *, int, int, int64_t *, double *, int *);
    int64_t (*corpus_rowid_at)(const vg_shards *, int64_t);
    int (*quantize_query)(int, const void *, int, float, float, int, void *);
    int (*corpus_minmax)(vg_shards *, float *, float *, int *);
    int (*corpus_quantize_rows)(vg_shards *, float, float, int, int64_t, int64_t, uint8_t *);
    int (*corpus_set_tie_order)(vg_shards *, int);
    int (*corpus_set_scan_filter)(vg_shards *, int);
    int (*corpus_rowids)(const vg_shards *, int64_t, int64_t, int64_t *);
    int64_t (*corpus_find_rowid)(const vg_shards *, int64_t);
    int (*corpus_patch_rows)(vg_shards *, const int64_t *, int64_t, const void *, int64_t);
    int (*corpus_delete_rows)(vg_shards *, const int64_t *, int64_t);
    char load_error[512];
} gpu_api;

static gpu_api G;           /* process-wide, filled once (the reference also keeps process globals: distance-cpu.c:20-21) */

static void *gpu_sym(const char *name) {
    void *p = dlsym(G.handle, name);
    if (!p && !G.load_error[0]) snprintf(G.load_error, sizeof(G.load_error), "libvectorgpu.so lacks symbol %s", name);
    return p;
}

static pthread_mutex_t gpu_load_lock = PTHREAD_MUTEX_INITIALIZER;
static int gpu_load_locked(void);

/* several connections (threads) may load the extension at once: the process-wide table is filled under a lock */
static int gpu_load(void) {
    pthread_mutex_lock(&gpu_load_lock);
    int ok = gpu_load_locked();
    pthread_mutex_unlock(&gpu_load_lock);
    return ok;
}

static int gpu_load_locked(void) {
    if (G.ready) return 1;
    if (G.load_error[0]) return 0;
    char path[PATH_MAX + 32];
    const char *env = getenv("VECTORGPU_LIB");
    Dl_info info;
    if (env && *env) {
        snprintf(path, sizeof(path), "%s", env);
    } else if (dladdr((void *)&gpu_load_locked, &info) && info.dli_fname) {
        snprintf(path, sizeof(path), "%s", info.dli_fname);
        char *slash = strrchr(path, '/');
        if (slash) slash[1] = 0; else path[0] = 0;
        strncat(path, "libvectorgpu.so", sizeof(path) - strlen(path) - 1);
    } else {
        snprintf(path, sizeof(path), "libvectorgpu.so");
    }
    G.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!G.handle) {
        snprintf(G.load_error, sizeof(G.load_error), "cannot load the GPU engine (%s): %s", path, dlerror());
        return 0;
    }
    G.device_count = (int (*)(void))gpu_sym("vg_device_count");
    G.backend_name = (const char *(*)(void))gpu_sym("vg_backend_name");
    G.last_error = (const char *(*)(void))gpu_sym("vg_last_error");
    G.corpus_create = (int (*)(const int *, int, int, int, int64_t, vg_shards **))gpu_sym("vg_shards_create");
    G.corpus_destroy = (void (*)(vg_shards *))gpu_sym("vg_shards_destroy");
    G.corpus_clear = (int (*)(vg_shards *))gpu_sym("vg_shards_clear");
    G.corpus_reserve = (int (*)(vg_shards *, int64_t))gpu_sym("vg_shards_reserve");
    G.corpus_rows = (int64_t (*)(const vg_shards *))gpu_sym("vg_shards_rows");
    G.corpus_append = (int (*)(vg_shards *, const void *, int64_t, int64_t, const int64_t *))gpu_sym("vg_shards_append");
    G.corpus_append_records = (int (*)(vg_shards *, const void *, int64_t))gpu_sym("vg_shards_append_records");
    G.scan_topk = (int (*)(vg_shards *, int, const void *, int, int64_t *, double *, int *))gpu_sym("vg_shards_scan_topk");
    G.scan_distances = (int (*)(vg_shards *, int, const void *, float *))gpu_sym("vg_shards_scan_distances");
    G.scan_topk_batch = (int (*)(vg_shards *, int, const void *, int, int, int64_t *, double *, int *))gpu_sym("vg_shards_scan_topk_batch");
    G.corpus_rowid_at = (int64_t (*)(const vg_shards *, int64_t))gpu_sym("vg_shards_rowid_at");
    G.quantize_query = (int (*)(int, const void *, int, float, float, int, void *))gpu_sym("vg_quantize_query");
    G.corpus_minmax = (int (*)(vg_shards *, float *, float *, int *))gpu_sym("vg_shards_minmax");
    G.corpus_quantize_rows = (int (*)(vg_shards *, float, float, int, int64_t, int64_t, uint8_t *))gpu_sym("vg_shards_quantize_rows");
    G.corpus_set_tie_order = (int (*)(vg_shards *, int))gpu_sym("vg_shards_set_tie_order");
    G.corpus_set_scan_filter = (int (*)(vg_shards *, int))gpu_sym("vg_shards_set_scan_filter");
    G.corpus_rowids = (int (*)(const vg_shards *, int64_t, int64_t, int64_t *))gpu_sym("vg_shards_rowids");
    G.corpus_find_rowid = (int64_t (*)(const vg_shards *, int64_t))gpu_sym("vg_shards_find_rowid");
    G.corpus_patch_rows = (int (*)(vg_shards *, const int64_t *, int64_t, const void *, int64_t))gpu_sym("vg_shards_patch_rows");
    G.corpus_delete_rows = (int (*)(vg_shards *, const int64_t *, int64_t))gpu_sym("vg_shards_delete_rows");
    if (G.load_error[0]) { dlclose(G.handle); G.handle = NULL; return 0; }
    G.ready = 1;
    return 1;
}

static const char *gpu_error(void) {
    if (!G.handle) return G.load_error[0] ? G.load_error : "GPU engine not loaded";
    return G.last_error();
}

/* Which devices hold a corpus.  Either the vector_init option gpu_devices=... (the reference ignores unknown option
 * keys, sqlite-vector.c:990-991, so a database initialised this way still opens there) or the environment variable
 * VECTORGPU_DEVICES: "all" | a count ("4" = devices 0..3) | a list ("0+2+5" in the option string, "0,2,5" or "0+2+5"
 * in the environment; a device may repeat).  Default: device 0.  More than one entry deals the rows block-cyclically
 * over the devices (gpu_shard_rows / VECTORGPU_SHARD_ROWS rows per block, default 65536) and every scan runs on all
 * of them at once (vg_shards). */
static int corpus_open_devices(const char *spec, int64_t shard_rows, int vtype, int dim, vg_shards **out) {
    int devs[64], n = 0;
    const char *e = (spec && *spec) ? spec : getenv("VECTORGPU_DEVICES");
    if (e && *e) {
        if (!strcasecmp(e, "all")) {
            n = G.device_count();
            if (n > 64) n = 64;
            for (int i = 0; i < n; ++i) devs[i] = i;
        } else if (strchr(e, ',') || strchr(e, '+')) {
            const char *p = e;
            while (*p && n < 64) {
                char *end;
                long v = strtol(p, &end, 10);
                if (end == p) break;
                devs[n++] = (int)v;
                p = (*end == ',' || *end == '+') ? end + 1 : end;
            }
        } else {
            n = atoi(e);
            if (n > 64) n = 64;
            for (int i = 0; i < n; ++i) devs[i] = i;
        }
    }
    if (n <= 0) { devs[0] = 0; n = 1; }
    const char *b = getenv("VECTORGPU_SHARD_ROWS");
    if (shard_rows <= 0 && b && *b) shard_rows = (int64_t)atoll(b);
    return G.corpus_create(devs, n, vtype, dim, shard_rows > 0 ? shard_rows : 0, out);
}

/* ------------------------------------------------------------------------------------------------ context */

typedef struct {
    int v_type;                 /* VG_TYPE_* (same numbering as the reference's vector_type) */
    int v_dim;
    int v_normalized;
    int v_distance;             /* VG_DIST_* */
    int q_type;                 /* VG_QUANT_* */
    uint64_t max_memory;
    char gpu_devices[64];       /* additions (ignored by the reference): where the corpus lives */
    int64_t gpu_shard_rows;
    int tie_order;              /* -1 = default, VG_TIE_POSITION, VG_TIE_REFERENCE (option tie_order=position|reference) */
    int scan_filter;            /* -1 = default, 0 / 1 (option scan_filter=0|1): f32 / f16 / bf16 scans through a shadow-copy filter */
    int track_changes;          /* -1 = default (VECTORGPU_TRACK_CHANGES, else off), 0 / 1 (option track_changes=0|1): row-granular
                                   freshness for UPDATE / DELETE through sqlite3_update_hook - see on_row_change() */
} vec_options;

/* Result order among EQUAL distances.  tie_order=reference is the reference's own result, rowid for rowid - its slot algorithm
 * (sqlite-vector.c:2022-2069, 2102-2106) is history dependent among ties.  It costs what tie_order=position costs unless the k + 1
 * best distances of a query hold a tie (then a host replay over the few rows that can enter the slots, vg_reforder.hip), so it is
 * the DEFAULT wherever ties are routine: integer element types - every vector_quantize_scan, and full scans of INT8 / UINT8
 * columns (north_star: "bit-exact rowid/top-k ordering for int8/uint8").  Float columns default to (distance, scan position):
 * their distances differ from the reference's in the last bits anyway (f32 <= 1e-5), exact ties are duplicates.
 * Explicit: the tie_order= option, else the VECTORGPU_TIE_ORDER environment variable. */
static int tie_order_for(const vec_options *o, int vtype) {
    if (o->tie_order >= 0) return o->tie_order;
    const char *e = getenv("VECTORGPU_TIE_ORDER");
    if (e && *e) return !strcasecmp(e, "reference") ? VG_TIE_REFERENCE : VG_TIE_POSITION;
    return (vtype == VG_TYPE_U8 || vtype == VG_TYPE_I8) ? VG_TIE_REFERENCE : VG_TIE_POSITION;
}

static int corpus_open_spec(const vec_options *o, int vtype, int dim, vg_shards **out) {
    int rc = corpus_open_devices(o->gpu_devices, o->gpu_shard_rows, vtype, dim, out);
    if (rc != VG_OK) return rc;
    if ((rc = G.corpus_set_tie_order(*out, tie_order_for(o, vtype))) != VG_OK ||
        (rc = G.corpus_set_scan_filter(*out, o->scan_filter)) != VG_OK) {
        G.corpus_destroy(*out);
        *out = NULL;
    }
    return rc;
}

typedef struct {
    char *t_name, *c_name, *pk_name;
    vec_options opt;
    float scale, offset;        /* quantization parameters (persisted as qscale / qoffset) */

    /* HBM-resident state owned by this (table, column) */
    vg_shards *full;            /* raw vectors for vector_full_scan[_stream] */
    int64_t full_data_version;  /* staleness stamps: PRAGMA data_version + sqlite3_total_changes() + PRAGMA schema_version */
    int64_t full_changes;
    int64_t full_schema;
    int64_t full_table_rows;    /* COUNT(*) of the table when it was staged (NULL vectors included) and its largest key: */
    int64_t full_max_pk;        /*   what the append-only check of stage_full() compares against */
    int full_have_pk;
    int full_in_txn;            /* staged inside an open transaction: a ROLLBACK leaves both stamps unchanged */
    int full_validated;         /* set when stage_full() (re)validated `full` during the current vector_quantize call */
    /* change tracking (track_changes=1): rowids of this table touched since `full` was last brought up to date */
    int64_t *touched;
    int n_touched, cap_touched, touched_overflow;
    int64_t hook_seen;          /* vec_context.hook_events at that moment; -1: `full` was staged without the hook in place */
    vg_shards *quant;           /* quantized vectors for vector_quantize_scan[_stream] */
    int quant_preloaded;        /* explicit vector_quantize_preload() (kept until cleanup / re-quantize) */
    int64_t quant_data_version;
    int64_t quant_changes;
    int64_t quant_schema;
    int quant_in_txn;
} table_ctx;

typedef struct {
    table_ctx tables[MAX_TABLES];
    int count;
    int hook_installed;         /* sqlite3_update_hook(db, on_row_change, this) was called for this connection */
    int64_t hook_events;        /* row changes it has reported so far (every table, every attached database) */
} vec_context;

static int elem_size(int t) {
    switch (t) {
        case VG_TYPE_F32: return 4;
        case VG_TYPE_F16: case VG_TYPE_BF16: return 2;
        case VG_TYPE_U8: case VG_TYPE_I8: return 1;
    }
    return 0;
}

static const char *type_name(int t) {
    switch (t) {
        case VG_TYPE_F32: return "FLOAT32"; case VG_TYPE_F16: return "FLOAT16"; case VG_TYPE_BF16: return "FLOATB16";
        case VG_TYPE_U8: return "UINT8"; case VG_TYPE_I8: return "INT8";
    }
    return "N/A";
}

static int type_from_name(const char *s) {
    if (!strcasecmp(s, "FLOAT32")) return VG_TYPE_F32;
    if (!strcasecmp(s, "FLOAT16")) return VG_TYPE_F16;
    if (!strcasecmp(s, "FLOATB16")) return VG_TYPE_BF16;
    if (!strcasecmp(s, "UINT8")) return VG_TYPE_U8;
    if (!strcasecmp(s, "INT8")) return VG_TYPE_I8;
    return 0;
}

static int distance_from_name(const char *s) {
    if (!strcasecmp(s, "L2") || !strcasecmp(s, "EUCLIDEAN")) return VG_DIST_L2;
    if (!strcasecmp(s, "SQUARED_L2")) return VG_DIST_SQUARED_L2;
    if (!strcasecmp(s, "COSINE")) return VG_DIST_COSINE;
    if (!strcasecmp(s, "DOT") || !strcasecmp(s, "INNER")) return VG_DIST_DOT;
    if (!strcasecmp(s, "L1") || !strcasecmp(s, "MANHATTAN")) return VG_DIST_L1;
    return 0;
}

static const char *sql_type_name(int t) {
    switch (t) {
        case SQLITE_TEXT: return "TEXT"; case SQLITE_INTEGER: return "INTEGER";
        case SQLITE_FLOAT: return "REAL"; case SQLITE_BLOB: return "BLOB";
    }
    return "N/A";
}

static char *dup_str(const char *s) {
    if (!s) return NULL;
    size_t n = strlen(s) + 1;
    char *r = (char *)sqlite3_malloc((int)n);
    if (r) memcpy(r, s, n);
    return r;
}

/* the message of the last ctx_error() on this thread: callers that post-process a failure (vector_quantize rolls back
 * and would otherwise report sqlite3_errmsg() = "not an error" for a failure that did not come from SQLite) keep it */
static __thread char last_ctx_error[1024];

static void ctx_error(sqlite3_context *ctx, int rc, const char *fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    strncpy(last_ctx_error, buf, sizeof(last_ctx_error) - 1);
    last_ctx_error[sizeof(last_ctx_error) - 1] = 0;
    sqlite3_result_error(ctx, buf, -1);
    sqlite3_result_error_code(ctx, rc);
}

static int vtab_error(sqlite3_vtab *vt, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    if (vt->zErrMsg) sqlite3_free(vt->zErrMsg);
    vt->zErrMsg = sqlite3_vmprintf(fmt, ap);
    va_end(ap);
    return SQLITE_ERROR;
}

static void table_release_gpu(table_ctx *t) {
    if (t->full) { G.corpus_destroy(t->full); t->full = NULL; }
    if (t->quant) { G.corpus_destroy(t->quant); t->quant = NULL; }
    t->quant_preloaded = 0;
}

static void context_free(void *p) {
    vec_context *c = (vec_context *)p;
    if (!c) return;
    for (int i = 0; i < c->count; ++i) {
        table_release_gpu(&c->tables[i]);
        sqlite3_free(c->tables[i].t_name);
        sqlite3_free(c->tables[i].c_name);
        sqlite3_free(c->tables[i].pk_name);
        sqlite3_free(c->tables[i].touched);
    }
    sqlite3_free(c);
}

static table_ctx *context_lookup(vec_context *c, const char *tbl, const char *col) {
    if (!tbl || !col) return NULL;
    for (int i = 0; i < c->count; ++i) {
        table_ctx *t = &c->tables[i];
        if (t->t_name && t->c_name && !strcasecmp(t->t_name, tbl) && !strcasecmp(t->c_name, col)) return t;
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------------ change tracking
 * The reference reads the table for every scan, so an UPDATE or DELETE costs it nothing extra.  A corpus resident in HBM has
 * to learn WHICH rows changed; SQLite tells exactly that to sqlite3_update_hook().  With track_changes=1 (vector_init option,
 * or VECTORGPU_TRACK_CHANGES=1) the extension installs the hook on the connection and logs the rowids it reports for a staged
 * table; the next scan re-reads only those rows and patches / removes / appends them on the device (stage_full).
 * Opt-in, because a connection has ONE update hook: installing ours replaces the application's (and theirs would replace
 * ours - which the event count below notices: the scan then falls back to the stamp logic).  The log only says where to look:
 * what a touched row IS now is read from the table, so rolled-back statements and re-inserted keys need no special case. */
#define TRACK_MAX_TOUCHED 65536
#define TRACK_MAX_DELETES 4096

static void on_row_change(void *p, int op, const char *dbname, const char *tbl, sqlite3_int64 rowid) {
    vec_context *vc = (vec_context *)p;
    (void)op;
    vc->hook_events++;                                          /* every report counts: compared with sqlite3_total_changes() */
    if (!tbl || !dbname || strcasecmp(dbname, "main")) return;
    for (int i = 0; i < vc->count; ++i) {
        table_ctx *t = &vc->tables[i];
        if (!t->full || t->hook_seen < 0 || t->touched_overflow || strcasecmp(t->t_name, tbl)) continue;
        if (t->n_touched > 0 && t->touched[t->n_touched - 1] == rowid) continue;
        if (t->n_touched == t->cap_touched) {
            int cap = t->cap_touched ? t->cap_touched * 2 : 256;
            int64_t *nb = (cap <= TRACK_MAX_TOUCHED) ? (int64_t *)sqlite3_realloc64(t->touched, (sqlite3_uint64)cap * sizeof(int64_t)) : NULL;
            if (!nb) { t->touched_overflow = 1; continue; }
            t->touched = nb; t->cap_touched = cap;
        }
        t->touched[t->n_touched++] = rowid;
    }
}

static int track_wanted(const vec_options *o) {
    if (o->track_changes >= 0) return o->track_changes;
    const char *e = getenv("VECTORGPU_TRACK_CHANGES");
    return e && *e && strcmp(e, "0") != 0;
}

static void track_install(sqlite3 *db, vec_context *vc) {
    if (vc->hook_installed) return;
    sqlite3_update_hook(db, on_row_change, vc);
    vc->hook_installed = 1;
}

/* `full` is up to date as of now: start a new log */
static void track_reset(vec_context *vc, table_ctx *t) {
    t->n_touched = 0;
    t->touched_overflow = 0;
    t->hook_seen = (vc && vc->hook_installed && track_wanted(&t->opt)) ? vc->hook_events : -1;
}

/* room for `need` rows of row_bytes each (geometric growth); 0 on allocation failure */
static int rows_grow(uint8_t **buf, int64_t *cap, int64_t need, int64_t row_bytes) {
    if (need <= *cap) return 1;
    int64_t ncap = *cap ? *cap * 2 : 64;
    if (ncap < need) ncap = need;
    uint8_t *nb = (uint8_t *)sqlite3_realloc64(*buf, (sqlite3_uint64)ncap * (sqlite3_uint64)row_bytes);
    if (!nb) return 0;
    *buf = nb; *cap = ncap;
    return 1;
}

static int cmp_i64(const void *a, const void *b) {
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------------------------------------ small SQL helpers */

static int64_t read_int64(sqlite3 *db, const char *sql) {
    sqlite3_stmt *st = NULL;
    int64_t v = 0;
    if (sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK && sqlite3_step(st) == SQLITE_ROW) v = sqlite3_column_int64(st, 0);
    sqlite3_finalize(st);
    return v;
}

static int exists_in_master(sqlite3 *db, const char *type, const char *name) {
    sqlite3_stmt *st = NULL;
    int found = 0;
    char sql[256];
    snprintf(sql, sizeof(sql), "SELECT EXISTS (SELECT 1 FROM sqlite_master WHERE type='%s' AND name=? COLLATE NOCASE);", type);
    if (sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK) {
        sqlite3_bind_text(st, 1, name, -1, SQLITE_STATIC);
        if (sqlite3_step(st) == SQLITE_ROW) found = sqlite3_column_int(st, 0);
    }
    sqlite3_finalize(st);
    return found;
}

static int column_exists(sqlite3 *db, const char *tbl, const char *col) {
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "SELECT EXISTS(SELECT 1 FROM pragma_table_info('%q') WHERE name = ?1);", tbl);
    sqlite3_stmt *st = NULL;
    int found = 0;
    if (sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK) {
        sqlite3_bind_text(st, 1, col, -1, SQLITE_STATIC);
        if (sqlite3_step(st) == SQLITE_ROW) found = sqlite3_column_int(st, 0) != 0;
    }
    sqlite3_finalize(st);
    return found;
}

static int column_is_blob(sqlite3 *db, const char *tbl, const char *col) {
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "SELECT type FROM pragma_table_info('%q') WHERE name=?", tbl);
    sqlite3_stmt *st = NULL;
    int ok = 0;
    if (sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK) {
        sqlite3_bind_text(st, 1, col, -1, SQLITE_STATIC);
        if (sqlite3_step(st) == SQLITE_ROW) {
            const char *decl = (const char *)sqlite3_column_text(st, 0);   /* BLOB affinity rule of datatype3.html */
            ok = (decl == NULL) || (strcasestr(decl, "BLOB") != NULL);
        }
    }
    sqlite3_finalize(st);
    return ok;
}

static int table_without_rowid(sqlite3 *db, const char *tbl) {
    sqlite3_stmt *st = NULL;
    int r = 0;
    if (sqlite3_prepare_v2(db, "SELECT sql FROM sqlite_master WHERE type='table' AND name=?", -1, &st, NULL) == SQLITE_OK) {
        sqlite3_bind_text(st, 1, tbl, -1, SQLITE_STATIC);
        if (sqlite3_step(st) == SQLITE_ROW) {
            const char *ddl = (const char *)sqlite3_column_text(st, 0);
            r = ddl && strcasestr(ddl, "WITHOUT ROWID");
        }
    }
    sqlite3_finalize(st);
    return r;
}

static char *single_int_pk(sqlite3 *db, const char *tbl) {
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "SELECT COUNT(*), type, name FROM pragma_table_info('%q') WHERE pk > 0;", tbl);
    sqlite3_stmt *st = NULL;
    char *pk = NULL;
    if (sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK && sqlite3_step(st) == SQLITE_ROW) {
        if (sqlite3_column_int(st, 0) == 1) {
            const char *decl = (const char *)sqlite3_column_text(st, 1);
            if (decl && strcasestr(decl, "INT")) pk = dup_str((const char *)sqlite3_column_text(st, 2));
        }
    }
    sqlite3_finalize(st);
    return pk;
}

static int meta_put(sqlite3_context *ctx, const char *tbl, const char *col, const char *key, int is_int, int64_t iv, double fv) {
    sqlite3 *db = sqlite3_context_db_handle(ctx);
    sqlite3_stmt *st = NULL;
    int rc = sqlite3_prepare_v2(db, "REPLACE INTO _sqliteai_vector (tblname, colname, key, value) VALUES (?, ?, ?, ?);", -1, &st, NULL);
    if (rc == SQLITE_OK) {
        sqlite3_bind_text(st, 1, tbl, -1, SQLITE_STATIC);
        sqlite3_bind_text(st, 2, col, -1, SQLITE_STATIC);
        sqlite3_bind_text(st, 3, key, -1, SQLITE_STATIC);
        if (is_int) sqlite3_bind_int64(st, 4, iv); else sqlite3_bind_double(st, 4, fv);
        rc = sqlite3_step(st);
        if (rc == SQLITE_DONE) rc = SQLITE_OK;
    }
    if (rc != SQLITE_OK) sqlite3_result_error(ctx, sqlite3_errmsg(db), -1);
    sqlite3_finalize(st);
    return rc;
}

/* re-read persisted qtype / qscale / qoffset (written by any build of the extension, reference included) */
static void meta_load(sqlite3 *db, table_ctx *t) {
    sqlite3_stmt *st = NULL;
    if (sqlite3_prepare_v2(db, "SELECT key, value FROM _sqliteai_vector WHERE tblname = ? AND colname = ?;", -1, &st, NULL) != SQLITE_OK) return;
    sqlite3_bind_text(st, 1, t->t_name, -1, SQLITE_STATIC);
    sqlite3_bind_text(st, 2, t->c_name, -1, SQLITE_STATIC);
    while (sqlite3_step(st) == SQLITE_ROW) {
        const char *key = (const char *)sqlite3_column_text(st, 0);
        if (!key) continue;
        if (!strcmp(key, "qtype")) t->opt.q_type = sqlite3_column_int(st, 1);
        else if (!strcmp(key, "qscale")) t->scale = (float)sqlite3_column_double(st, 1);
        else if (!strcmp(key, "qoffset")) t->offset = (float)sqlite3_column_double(st, 1);
    }
    sqlite3_finalize(st);
}

/* ------------------------------------------------------------------------------------------------ option strings */

static void options_default(vec_options *o) {
    memset(o, 0, sizeof(*o));
    o->v_type = VG_TYPE_F32;
    o->v_distance = VG_DIST_L2;
    o->max_memory = DEFAULT_MAX_MEMORY;
    o->q_type = VG_QUANT_AUTO;
    o->tie_order = -1;
    o->scan_filter = -1;
    o->track_changes = -1;
}

static uint64_t parse_size(const char *s) {
    char *end = NULL;
    double d = strtod(s, &end);
    if (d == 0 || d == HUGE_VAL) return 0;
    while (*end && isspace((unsigned char)*end)) end++;
    if (!strncasecmp(end, "KB", 2)) d *= 1024.0;
    else if (!strncasecmp(end, "MB", 2)) d *= 1024.0 * 1024.0;
    else if (!strncasecmp(end, "GB", 2)) d *= 1024.0 * 1024.0 * 1024.0;
    else if (*end) return 0;
    if (d < 0 || d > (double)INT64_MAX) return 0;
    return (uint64_t)d;
}

/* one key=value pair; keys are matched on the typed prefix exactly like the reference (strncasecmp with the key's
 * own length, sqlite-vector.c:950-983).  Returns 0 after raising an SQL error. */
static int option_apply(sqlite3_context *ctx, vec_options *o, const char *key, int klen, const char *val, int vlen) {
    if (klen <= 0 || vlen <= 0) return 0;
    char v[256] = {0};
    memcpy(v, val, vlen > 255 ? 255 : (size_t)vlen);
    if (!strncasecmp(key, "type", (size_t)klen)) {
        int t = type_from_name(v);
        if (!t) { ctx_error(ctx, SQLITE_ERROR, "Invalid vector type: '%s' is not a recognized type.", v); return 0; }
        o->v_type = t;
    } else if (!strncasecmp(key, "dimension", (size_t)klen)) {
        int d = (int)strtol(v, NULL, 0);
        if (d <= 0) { ctx_error(ctx, SQLITE_ERROR, "Invalid vector dimension: expected a positive integer, got '%s'.", v); return 0; }
        o->v_dim = d;
    } else if (!strncasecmp(key, "normalized", (size_t)klen)) {
        o->v_normalized = strtol(v, NULL, 0) != 0;
    } else if (!strncasecmp(key, "max_memory", (size_t)klen)) {
        o->max_memory = (uint64_t)(int)parse_size(v);          /* the reference truncates through int (:972) */
    } else if (!strncasecmp(key, "qtype", (size_t)klen)) {
        if (!strcasecmp(v, "UINT8")) o->q_type = VG_QUANT_U8;
        else if (!strcasecmp(v, "INT8")) o->q_type = VG_QUANT_S8;
        else { ctx_error(ctx, SQLITE_ERROR, "Invalid quantization type: '%s' is not a recognized or supported quantization type.", v); return 0; }
    } else if (!strncasecmp(key, "distance", (size_t)klen)) {
        int d = distance_from_name(v);
        if (!d) { ctx_error(ctx, SQLITE_ERROR, "Invalid distance name: '%s' is not a recognized or supported distance.", v); return 0; }
        o->v_distance = d;
    } else if (!strncasecmp(key, "gpu_devices", (size_t)klen) && klen == 11) {
        snprintf(o->gpu_devices, sizeof(o->gpu_devices), "%s", v);
    } else if (!strncasecmp(key, "gpu_shard_rows", (size_t)klen) && klen == 14) {
        o->gpu_shard_rows = (int64_t)strtoll(v, NULL, 0);
    } else if (!strncasecmp(key, "tie_order", (size_t)klen) && klen == 9) {
        if (!strcasecmp(v, "reference")) o->tie_order = VG_TIE_REFERENCE;
        else if (!strcasecmp(v, "position")) o->tie_order = VG_TIE_POSITION;
        else { ctx_error(ctx, SQLITE_ERROR, "Invalid tie_order: '%s' (expected 'reference' or 'position').", v); return 0; }
    } else if (!strncasecmp(key, "scan_filter", (size_t)klen) && klen == 11) {
        o->scan_filter = strtol(v, NULL, 0) != 0;
    } else if (!strncasecmp(key, "track_changes", (size_t)klen) && klen == 13) {
        o->track_changes = strtol(v, NULL, 0) != 0;
    }
    return 1;                                                   /* unknown keys are ignored */
}

static int options_parse(sqlite3_context *ctx, const char *s, vec_options *o) {
    if (!s) return 1;
    const char *p = s;
    while (*p) {
        while (*p && isspace((unsigned char)*p)) p++;
        const char *k0 = p;
        while (*p && *p != '=' && *p != ',') p++;
        int klen = (int)(p - k0);
        while (klen > 0 && isspace((unsigned char)k0[klen - 1])) klen--;
        if (*p != '=') {                                        /* malformed pair: skip it */
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
            continue;
        }
        p++;
        while (*p && isspace((unsigned char)*p)) p++;
        const char *v0 = p;
        while (*p && *p != ',') p++;
        int vlen = (int)(p - v0);
        while (vlen > 0 && isspace((unsigned char)v0[vlen - 1])) vlen--;
        if (!option_apply(ctx, o, k0, klen, v0, vlen)) return 0;
        if (*p == ',') p++;
    }
    return 1;
}

static int check_args(sqlite3_context *ctx, const char *fn, int argc, sqlite3_value **argv, int n, const int *types) {
    if (argc != n) { ctx_error(ctx, SQLITE_ERROR, "Function '%s' expects %d arguments, but %d were provided.", fn, n, argc); return 0; }
    for (int i = 0; i < argc; ++i) {
        int t = sqlite3_value_type(argv[i]);
        if (t != types[i]) {
            ctx_error(ctx, SQLITE_ERROR, "Function '%s': argument %d must be of type %s (got %s).", fn, i + 1, sql_type_name(types[i]), sql_type_name(t));
            return 0;
        }
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------ 16-bit conversions */

static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, out;
    float f;
    if (e == 0x1F) out = sign | 0x7F800000u | (m << 13);
    else if (e) out = sign | ((e + 112u) << 23) | (m << 13);
    else if (!m) out = sign;
    else { f = (float)m * 0x1.0p-24f; memcpy(&out, &f, 4); out |= sign; }
    memcpy(&f, &out, 4);
    return f;
}

static uint16_t f32_to_f16(float f) {            /* round to nearest even */
    uint32_t w; memcpy(&w, &f, 4);
    uint16_t sign = (uint16_t)((w >> 16) & 0x8000u);
    uint32_t a = w & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);
    if (a < 0x33000000u) return sign;
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t kept = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (kept & 1u))) kept++;
    uint32_t out = (e < -14) ? kept : (((uint32_t)(e + 15) << 10) + (kept - 0x400u));
    if (out >= 0x7C00u) out = 0x7C00u;
    return (uint16_t)(sign | out);
}

static float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f32_to_bf16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    return (uint16_t)((x + 0x7FFFu + ((x >> 16) & 1u)) >> 16);
}

static float elem_as_f32(int type, const void *blob, int i) {
    switch (type) {
        case VG_TYPE_F32: return ((const float *)blob)[i];
        case VG_TYPE_F16: return f16_to_f32(((const uint16_t *)blob)[i]);
        case VG_TYPE_BF16: return bf16_to_f32(((const uint16_t *)blob)[i]);
        case VG_TYPE_U8: return (float)((const uint8_t *)blob)[i];
        case VG_TYPE_I8: return (float)((const int8_t *)blob)[i];
    }
    return 0.0f;
}

/* ------------------------------------------------------------------------------------------------ JSON -> BLOB */

/* "[1, 2.5, ...]" -> vector of `type`; error goes to ctx or vtab.  Returns sqlite3_malloc'd buffer. */
static void *vector_from_json(sqlite3_context *ctx, sqlite3_vtab *vt, int type, const char *json, int *size, int dim) {
#define JSON_FAIL(...) do { if (vt) vtab_error(vt, __VA_ARGS__); else if (ctx) ctx_error(ctx, SQLITE_ERROR, __VA_ARGS__); sqlite3_free(blob); return NULL; } while (0)
    char *blob = NULL;
    while (*json && isspace((unsigned char)*json)) json++;
    if (*json != '[') JSON_FAIL("Malformed JSON: expected '[' at the beginning of the array.");
    json++;
    int commas = 0;
    for (const char *p = json; *p; ++p) if (*p == ',') commas++;
    int es = elem_size(type), cap = commas + 1, count = 0;
    blob = (char *)sqlite3_malloc(cap * es);
    if (!blob) JSON_FAIL("Out of memory: unable to allocate %lld bytes for BLOB buffer.", (long long)cap * es);
    const char *p = json;
    while (*p) {
        while (*p && isspace((unsigned char)*p)) p++;
        if (*p == ']') break;
        char *end;
        double v = strtod(p, &end);
        if (end == p) JSON_FAIL("Malformed JSON: expected a number at position %d (found '%c').", (int)(p - json) + 1, *p ? *p : '?');
        if (count >= cap) JSON_FAIL("Too many elements in JSON array.");
        switch (type) {
            case VG_TYPE_F32: ((float *)blob)[count++] = (float)v; break;
            case VG_TYPE_F16: ((uint16_t *)blob)[count++] = f32_to_f16((float)v); break;
            case VG_TYPE_BF16: ((uint16_t *)blob)[count++] = f32_to_bf16((float)v); break;
            case VG_TYPE_U8:
                if (v < 0 || v > 255) JSON_FAIL("Value out of range for uint8_t.");
                ((uint8_t *)blob)[count++] = (uint8_t)v; break;
            case VG_TYPE_I8:
                if (v < -128 || v > 127) JSON_FAIL("Value out of range for int8_t.");
                ((int8_t *)blob)[count++] = (int8_t)v; break;
            default: JSON_FAIL("Unsupported vector type.");
        }
        p = end;
        while (*p && isspace((unsigned char)*p)) p++;
        if (*p == ',') {
            p++;
            while (*p && isspace((unsigned char)*p)) p++;
            if (*p == ']') break;                              /* trailing comma tolerated */
        } else if (*p == ']') {
            break;
        } else {
            JSON_FAIL("Malformed JSON: unexpected character '%c' at position %d.", *p ? *p : '?', (int)(p - json) + 1);
        }
    }
    if (dim > 0 && dim != count) JSON_FAIL("Invalid JSON vector dimension: expected %d but found %d.", dim, count);
    if (size) *size = count * es;
    return blob;
#undef JSON_FAIL
}

static void vector_as_type(sqlite3_context *ctx, int type, int argc, sqlite3_value **argv) {
    sqlite3_value *v = argv[0];
    int vbytes = sqlite3_value_bytes(v), vtype = sqlite3_value_type(v);
    int dim = (argc == 2) ? sqlite3_value_int(argv[1]) : 0;
    int es = elem_size(type);
    if (vtype == SQLITE_BLOB) {
        if (vbytes % es) { ctx_error(ctx, SQLITE_ERROR, "Invalid BLOB size for format '%s': size must be a multiple of %d bytes.", type_name(type), es); return; }
        if (dim > 0 && vbytes != es * dim) {
            ctx_error(ctx, SQLITE_ERROR, "Invalid BLOB size for format '%s': expected dimension should be %d (BLOB is %d bytes instead of %d).", type_name(type), dim, vbytes, es * dim);
            return;
        }
        sqlite3_result_value(ctx, v);
        return;
    }
    if (vtype == SQLITE_TEXT) {
        const char *json = (const char *)sqlite3_value_text(v);
        if (!json) { ctx_error(ctx, SQLITE_ERROR, "Invalid TEXT input."); return; }
        int size = 0;
        void *blob = vector_from_json(ctx, NULL, type, json, &size, dim);
        if (blob) sqlite3_result_blob(ctx, blob, size, sqlite3_free);
        return;
    }
    ctx_error(ctx, SQLITE_ERROR, "Unsupported input type: only BLOB and TEXT values are accepted (received %s).", sql_type_name(vtype));
}

static void fn_as_f32(sqlite3_context *c, int n, sqlite3_value **v) { vector_as_type(c, VG_TYPE_F32, n, v); }
static void fn_as_f16(sqlite3_context *c, int n, sqlite3_value **v) { vector_as_type(c, VG_TYPE_F16, n, v); }
static void fn_as_bf16(sqlite3_context *c, int n, sqlite3_value **v) { vector_as_type(c, VG_TYPE_BF16, n, v); }
static void fn_as_i8(sqlite3_context *c, int n, sqlite3_value **v) { vector_as_type(c, VG_TYPE_I8, n, v); }
static void fn_as_u8(sqlite3_context *c, int n, sqlite3_value **v) { vector_as_type(c, VG_TYPE_U8, n, v); }

/* ------------------------------------------------------------------------------------------------ staging into HBM */

static void db_stamps(sqlite3 *db, int64_t *data_version, int64_t *changes, int64_t *schema) {
    *data_version = read_int64(db, "PRAGMA data_version;");      /* bumps when ANOTHER connection commits */
    *changes = (int64_t)sqlite3_total_changes(db);                /* bumps when THIS connection writes rows */
    *schema = read_int64(db, "PRAGMA schema_version;");          /* bumps on DROP / CREATE / ALTER (no row change is counted
                                                                     for DROP TABLE t; CREATE TABLE t ...) */
}

/* rows of the staging loop shared by the full and the append-only pass */
static int stage_rows(sqlite3 *db, table_ctx *t, sqlite3_stmt *st, char **err) {
    const int es = elem_size(t->opt.v_type), dim = t->opt.v_dim;
    const int64_t row_bytes = (int64_t)es * dim;
    uint8_t *stage = (uint8_t *)sqlite3_malloc64((sqlite3_uint64)STAGE_ROWS * (sqlite3_uint64)row_bytes);
    int64_t *ids = (int64_t *)sqlite3_malloc64((sqlite3_uint64)STAGE_ROWS * sizeof(int64_t));
    if (!stage || !ids) { sqlite3_free(stage); sqlite3_free(ids); return SQLITE_NOMEM; }
    int fill = 0, rc;
    while (1) {
        rc = sqlite3_step(st);
        if (rc != SQLITE_ROW) break;
        if (sqlite3_column_type(st, 1) == SQLITE_NULL) continue;
        const void *blob = sqlite3_column_blob(st, 1);
        if (!blob) continue;
        if (sqlite3_column_bytes(st, 1) < row_bytes) {
            *err = sqlite3_mprintf("Invalid vector blob found at rowid %lld.", (long long)sqlite3_column_int64(st, 0));
            rc = SQLITE_ERROR;
            break;
        }
        memcpy(stage + (int64_t)fill * row_bytes, blob, (size_t)row_bytes);
        ids[fill++] = sqlite3_column_int64(st, 0);
        if (fill == STAGE_ROWS) {
            if (G.corpus_append(t->full, stage, fill, row_bytes, ids) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); rc = SQLITE_ERROR; break; }
            fill = 0;
        }
    }
    if (rc == SQLITE_DONE) {
        rc = SQLITE_OK;
        if (fill && G.corpus_append(t->full, stage, fill, row_bytes, ids) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); rc = SQLITE_ERROR; }
    } else if (rc != SQLITE_ERROR || !*err) {
        if (!*err) *err = sqlite3_mprintf("%s", sqlite3_errmsg(db));
    }
    sqlite3_free(stage);
    sqlite3_free(ids);
    return rc;
}

/* COUNT(*) and MAX(pk) of the table as staged: the reference points of the append-only check */
static void table_watermark(sqlite3 *db, table_ctx *t) {
    char *sql = sqlite3_mprintf("SELECT COUNT(*), MAX(%q) FROM %q;", t->pk_name, t->t_name);
    sqlite3_stmt *st = NULL;
    t->full_table_rows = 0; t->full_max_pk = 0; t->full_have_pk = 0;
    if (sql && sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK && sqlite3_step(st) == SQLITE_ROW) {
        t->full_table_rows = sqlite3_column_int64(st, 0);
        if (sqlite3_column_type(st, 1) == SQLITE_INTEGER) { t->full_max_pk = sqlite3_column_int64(st, 1); t->full_have_pk = 1; }
    }
    sqlite3_finalize(st);
    sqlite3_free(sql);
}

/* Only this connection wrote since the corpus was staged (data_version and schema_version are unchanged), and
 * sqlite3_total_changes() grew by d.  If the table now holds exactly d more rows and exactly d rows lie above the key
 * watermark, then all d changes were INSERTs into THIS table behind every staged row (an UPDATE or DELETE anywhere, or
 * an INSERT elsewhere, raises the change counter without raising both counts): the staged rows are still the table's
 * rows in scan order and only the d new ones have to go to the device.  The scan order must be the key order for this
 * (a covering index on the vector column would make "SELECT pk, col" walk the index instead): checked with the query
 * plan.  Returns 1 and the statement that yields the new rows, else 0 (caller re-stages everything). */
static int append_only_since_staged(sqlite3 *db, table_ctx *t, int64_t d, sqlite3_stmt **new_rows) {
    *new_rows = NULL;
    if (d <= 0 || !t->full_have_pk || getenv("VECTORGPU_NO_INCREMENTAL")) return 0;
    int ok = 0;
    char *sql = sqlite3_mprintf("SELECT (SELECT COUNT(*) FROM %q), (SELECT COUNT(*) FROM %q WHERE %q > %lld);", t->t_name, t->t_name,
                                t->pk_name, (long long)t->full_max_pk);
    sqlite3_stmt *st = NULL;
    if (sql && sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK && sqlite3_step(st) == SQLITE_ROW)
        ok = (sqlite3_column_int64(st, 0) - t->full_table_rows == d) && (sqlite3_column_int64(st, 1) == d);
    sqlite3_finalize(st);
    sqlite3_free(sql);
    if (!ok) return 0;
    sql = sqlite3_mprintf("EXPLAIN QUERY PLAN SELECT %q, %q FROM %q;", t->pk_name, t->c_name, t->t_name);
    st = NULL;
    if (sql && sqlite3_prepare_v2(db, sql, -1, &st, NULL) == SQLITE_OK) {
        while (sqlite3_step(st) == SQLITE_ROW) {
            const char *detail = (const char *)sqlite3_column_text(st, 3);
            if (detail && strstr(detail, "INDEX")) ok = 0;               /* "SCAN t USING COVERING INDEX ..." */
        }
    } else ok = 0;
    sqlite3_finalize(st);
    sqlite3_free(sql);
    if (!ok) return 0;
    sql = sqlite3_mprintf("SELECT %q, %q FROM %q WHERE %q > %lld ORDER BY %q;", t->pk_name, t->c_name, t->t_name, t->pk_name,
                          (long long)t->full_max_pk, t->pk_name);
    if (!sql || sqlite3_prepare_v2(db, sql, -1, new_rows, NULL) != SQLITE_OK) { sqlite3_free(sql); *new_rows = NULL; return 0; }
    sqlite3_free(sql);
    return 1;
}

/* Bring `full` up to date from the change log (see on_row_change): every touched rowid is looked up in the table NOW and in
 * the corpus; present in both = patch, in the corpus only = remove, in the table only = append (possible only behind every
 * staged row).  Then COUNT(col) must equal the corpus' row count - what the hook cannot see (rows removed by ON CONFLICT
 * REPLACE, a key UPDATE's old rowid) shows up there.  Returns 1 when `full` is current, 0 when the caller has to re-stage
 * (nothing is left half-applied that a re-stage would not overwrite), -1 with *err on a hard error. */
static int apply_tracked_changes(sqlite3 *db, table_ctx *t, char **err) {
    const int es = elem_size(t->opt.v_type), dim = t->opt.v_dim;
    const int64_t row_bytes = (int64_t)es * dim;
    if (t->n_touched == 0) return 1;
    qsort(t->touched, (size_t)t->n_touched, sizeof(int64_t), cmp_i64);
    int n = 0;
    for (int i = 0; i < t->n_touched; ++i) if (n == 0 || t->touched[n - 1] != t->touched[i]) t->touched[n++] = t->touched[i];
    int result = 0, npatch = 0, ndel = 0, napp = 0;
    int64_t *ppos = (int64_t *)sqlite3_malloc64((sqlite3_uint64)n * sizeof(int64_t));
    int64_t *dpos = (int64_t *)sqlite3_malloc64((sqlite3_uint64)n * sizeof(int64_t));
    int64_t *aids = (int64_t *)sqlite3_malloc64((sqlite3_uint64)n * sizeof(int64_t));
    uint8_t *pdata = NULL, *adata = NULL;                       /* the touched rows' vectors: grown as they are met, not n rows up front */
    int64_t pcap = 0, acap = 0;
    sqlite3_stmt *st = NULL;
    char *sql = sqlite3_mprintf("SELECT %q FROM %q WHERE %q = ?1;", t->c_name, t->t_name, t->pk_name);
    if (!ppos || !dpos || !aids || !sql || sqlite3_prepare_v2(db, sql, -1, &st, NULL) != SQLITE_OK) goto done;
    {
        const int64_t rows0 = G.corpus_rows(t->full);
        int64_t last_id = rows0 > 0 ? G.corpus_rowid_at(t->full, rows0 - 1) : INT64_MIN;
        for (int i = 0; i < n; ++i) {
            const int64_t r = t->touched[i];
            const int64_t pos = G.corpus_find_rowid(t->full, r);
            if (pos == -2) goto done;                            /* several shards, or rowids not in key order */
            sqlite3_reset(st);
            sqlite3_bind_int64(st, 1, r);
            const void *blob = NULL;
            int rc = sqlite3_step(st);
            if (rc == SQLITE_ROW && sqlite3_column_type(st, 0) != SQLITE_NULL) {
                blob = sqlite3_column_blob(st, 0);
                if (blob && sqlite3_column_bytes(st, 0) < row_bytes) goto done;      /* (the full pass reports the short BLOB) */
            } else if (rc != SQLITE_ROW && rc != SQLITE_DONE) goto done;
            if (pos >= 0 && blob) {
                if (!rows_grow(&pdata, &pcap, npatch + 1, row_bytes)) goto done;
                ppos[npatch] = pos; memcpy(pdata + (int64_t)npatch * row_bytes, blob, (size_t)row_bytes); ++npatch;
            } else if (pos >= 0) { dpos[ndel++] = pos; }
            else if (blob) {
                if (r <= last_id) goto done;                     /* a new row in the middle of the scan order */
                if (!rows_grow(&adata, &acap, napp + 1, row_bytes)) goto done;
                aids[napp] = r; memcpy(adata + (int64_t)napp * row_bytes, blob, (size_t)row_bytes); ++napp;
                last_id = r;
            }
        }
    }
    if (ndel > TRACK_MAX_DELETES) goto done;
    /* positions are pre-deletion indices: patches first, then the removals (ascending: rowids ascend with positions), then appends */
    if (npatch && G.corpus_patch_rows(t->full, ppos, npatch, pdata, row_bytes) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); result = -1; goto done; }
    if (ndel && G.corpus_delete_rows(t->full, dpos, ndel) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); result = -1; goto done; }
    if (napp && G.corpus_append(t->full, adata, napp, row_bytes, aids) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); result = -1; goto done; }
    {
        char *cnt = sqlite3_mprintf("SELECT COUNT(%q) FROM %q;", t->c_name, t->t_name);
        const int64_t have = cnt ? read_int64(db, cnt) : -1;
        sqlite3_free(cnt);
        result = (have == G.corpus_rows(t->full)) ? 1 : 0;
    }
done:
    sqlite3_finalize(st);
    sqlite3_free(sql);
    sqlite3_free(ppos); sqlite3_free(dpos); sqlite3_free(aids); sqlite3_free(pdata); sqlite3_free(adata);
    return result;
}

/* Stage (or re-stage, if the database changed since) the raw vectors of (table, column) into HBM in the order the
 * reference scans them: "SELECT pk, col FROM tbl" (sqlite-vector.c:2077), NULL vectors skipped (:2093).
 * Short BLOBs are an error here (the reference would read past them, :2095-2098). */
static int stage_full(sqlite3 *db, vec_context *vc, table_ctx *t, char **err) {
    int64_t dv, ch, sv;
    db_stamps(db, &dv, &ch, &sv);
    /* rows staged inside an open transaction may be rolled back without either stamp moving (total_changes never
     * decreases): such a copy is good for one scan only - which is what the reference does for every scan anyway */
    if (t->full && !t->full_in_txn && t->full_data_version == dv && t->full_changes == ch && t->full_schema == sv) return SQLITE_OK;
    if (!gpu_load()) { *err = sqlite3_mprintf("%s", gpu_error()); return SQLITE_ERROR; }
    const int dim = t->opt.v_dim;
    sqlite3_stmt *st = NULL;
    int rc;
    /* row-granular freshness, any statement (track_changes=1): only this connection wrote (data_version, schema_version
     * unchanged) and the update hook reported exactly as many row changes as sqlite3_total_changes() counted - none escaped
     * it (WITHOUT ROWID tables, the truncate optimisation of DELETE without WHERE, a hook replaced by the application) */
    if (t->full && !t->full_in_txn && t->full_data_version == dv && t->full_schema == sv && vc && vc->hook_installed &&
        t->hook_seen >= 0 && !t->touched_overflow && track_wanted(&t->opt) && !getenv("VECTORGPU_NO_INCREMENTAL") &&
        vc->hook_events - t->hook_seen == ch - t->full_changes) {
        const int had = t->n_touched;
        const int r = apply_tracked_changes(db, t, err);
        if (r < 0) { G.corpus_destroy(t->full); t->full = NULL; return SQLITE_ERROR; }
        if (r > 0) {
            t->full_changes = ch;
            t->full_in_txn = !sqlite3_get_autocommit(db);
            if (had) table_watermark(db, t);
            track_reset(vc, t);
            return SQLITE_OK;
        }
    }
    if (t->full && !t->full_in_txn && t->full_data_version == dv && t->full_schema == sv &&
        append_only_since_staged(db, t, ch - t->full_changes, &st)) {
        /* row-granular freshness: the new rows are appended behind the staged ones (the device extends its cached
         * per-row data - norms, shadow copies - for the appended rows only) */
        rc = stage_rows(db, t, st, err);
        sqlite3_finalize(st);
        if (rc == SQLITE_OK) {
            t->full_changes = ch;
            t->full_in_txn = !sqlite3_get_autocommit(db);
            table_watermark(db, t);
            track_reset(vc, t);
        } else { G.corpus_destroy(t->full); t->full = NULL; }
        return rc;
    }
    if (t->full) G.corpus_clear(t->full);
    else if (corpus_open_spec(&t->opt, t->opt.v_type, dim, &t->full) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); return SQLITE_ERROR; }

    {   /* one HBM allocation of the right size instead of geometric regrowth (COUNT(*) is an upper bound: NULLs) */
        char *cnt = sqlite3_mprintf("SELECT COUNT(*) FROM %q;", t->t_name);
        if (cnt) { int64_t n = read_int64(db, cnt); sqlite3_free(cnt); if (n > 0) G.corpus_reserve(t->full, n); }
    }
    char *sql = sqlite3_mprintf("SELECT %q, %q FROM %q;", t->pk_name, t->c_name, t->t_name);
    if (!sql) return SQLITE_NOMEM;
    rc = sqlite3_prepare_v2(db, sql, -1, &st, NULL);
    sqlite3_free(sql);
    if (rc != SQLITE_OK) { *err = sqlite3_mprintf("%s", sqlite3_errmsg(db)); return rc; }
    rc = stage_rows(db, t, st, err);
    sqlite3_finalize(st);
    if (rc == SQLITE_OK) {
        t->full_data_version = dv; t->full_changes = ch; t->full_schema = sv; t->full_in_txn = !sqlite3_get_autocommit(db);
        table_watermark(db, t);
        track_reset(vc, t);
    } else { G.corpus_destroy(t->full); t->full = NULL; }
    return rc;
}

/* Stage the persisted quantized records (vector0_<tbl>_<col>.data = counter x [int64 LE rowid | dim bytes],
 * sqlite-vector.c:1296-1309) into HBM; this is what vector_quantize_preload does with a malloc'd buffer (:1338-1404). */
static int stage_quant(sqlite3 *db, table_ctx *t, int force, char **err) {
    int64_t dv, ch, sv;
    db_stamps(db, &dv, &ch, &sv);
    if (!force && t->quant && !t->quant_in_txn && t->quant_data_version == dv && t->quant_changes == ch && t->quant_schema == sv) return SQLITE_OK;
    if (!gpu_load()) { *err = sqlite3_mprintf("%s", gpu_error()); return SQLITE_ERROR; }
    const int vt = (t->opt.q_type == VG_QUANT_U8) ? VG_TYPE_U8 : VG_TYPE_I8;
    if (t->quant) { G.corpus_destroy(t->quant); t->quant = NULL; }
    if (corpus_open_spec(&t->opt, vt, t->opt.v_dim, &t->quant) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); return SQLITE_ERROR; }
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "SELECT counter, data FROM vector0_%q_%q;", t->t_name, t->c_name);
    sqlite3_stmt *st = NULL;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, NULL);
    if (rc != SQLITE_OK) { *err = sqlite3_mprintf("%s", sqlite3_errmsg(db)); G.corpus_destroy(t->quant); t->quant = NULL; return rc; }
    const int64_t rec = 8 + (int64_t)t->opt.v_dim;
    while ((rc = sqlite3_step(st)) == SQLITE_ROW) {
        int64_t n = sqlite3_column_int64(st, 0);
        const void *data = sqlite3_column_blob(st, 1);
        int64_t bytes = sqlite3_column_bytes(st, 1);
        if (!data || n <= 0) continue;
        if (bytes < n * rec) { *err = sqlite3_mprintf("Corrupt quantization chunk (%lld bytes for %lld records).", (long long)bytes, (long long)n); rc = SQLITE_ERROR; break; }
        if (G.corpus_append_records(t->quant, data, n) != VG_OK) { *err = sqlite3_mprintf("%s", gpu_error()); rc = SQLITE_ERROR; break; }
    }
    sqlite3_finalize(st);
    if (rc == SQLITE_DONE) rc = SQLITE_OK;
    if (rc == SQLITE_OK) { t->quant_data_version = dv; t->quant_changes = ch; t->quant_schema = sv; t->quant_in_txn = !sqlite3_get_autocommit(db); }
    else { G.corpus_destroy(t->quant); t->quant = NULL; }
    return rc;
}

/* ------------------------------------------------------------------------------------------------ vector_init */

static void fn_vector_init(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    static const int types[] = {SQLITE_TEXT, SQLITE_TEXT, SQLITE_TEXT};
    if (!check_args(ctx, "vector_init", argc, argv, 3, types)) return;
    const char *tbl = (const char *)sqlite3_value_text(argv[0]);
    const char *col = (const char *)sqlite3_value_text(argv[1]);
    const char *opts = (const char *)sqlite3_value_text(argv[2]);
    sqlite3 *db = sqlite3_context_db_handle(ctx);
    if (!exists_in_master(db, "table", tbl)) { ctx_error(ctx, SQLITE_ERROR, "Table '%s' does not exist.", tbl); return; }
    if (!column_exists(db, tbl, col)) { ctx_error(ctx, SQLITE_ERROR, "Column '%s' does not exist in table '%s'.", col, tbl); return; }
    if (!column_is_blob(db, tbl, col)) { ctx_error(ctx, SQLITE_ERROR, "Column '%s' in table '%s' must be of type BLOB.", col, tbl); return; }
    vec_options o;
    options_default(&o);
    if (!options_parse(ctx, opts, &o)) return;
    if (o.v_type == 0) { ctx_error(ctx, SQLITE_ERROR, "Vector type value is mandatory in vector_init"); return; }
    if (o.v_dim == 0) { ctx_error(ctx, SQLITE_ERROR, "Vector dimension value is mandatory in vector_init"); return; }

    vec_context *vc = (vec_context *)sqlite3_user_data(ctx);
    table_ctx *t = context_lookup(vc, tbl, col);
    if (t) {
        if (o.v_dim != t->opt.v_dim) { ctx_error(ctx, SQLITE_ERROR, "Inconsistent vector dimension for '%s.%s': existing=%d, provided=%d.", tbl, col, t->opt.v_dim, o.v_dim); return; }
        if (o.v_type != t->opt.v_type) { ctx_error(ctx, SQLITE_ERROR, "Inconsistent vector type for '%s.%s': existing=%s, provided=%s.", tbl, col, type_name(t->opt.v_type), type_name(o.v_type)); return; }
        if (o.v_normalized != t->opt.v_normalized) { ctx_error(ctx, SQLITE_ERROR, "Inconsistent normalization flag for '%s.%s': existing=%s, provided=%s.", tbl, col, t->opt.v_normalized ? "true" : "false", o.v_normalized ? "true" : "false"); return; }
        /* the GPU knobs (ignored by the reference) may be changed by calling vector_init again: they apply at once */
        if (o.tie_order >= 0) t->opt.tie_order = o.tie_order;
        if (o.scan_filter >= 0) t->opt.scan_filter = o.scan_filter;
        if (o.track_changes >= 0) t->opt.track_changes = o.track_changes;
        if (track_wanted(&t->opt)) track_install(db, vc);
        if ((o.tie_order >= 0 || o.scan_filter >= 0) && G.ready) {
            vg_shards *hs[2] = {t->full, t->quant};
            for (int i = 0; i < 2; ++i) {
                if (!hs[i]) continue;
                /* t->quant always holds uint8 / int8 records; t->full the column's own type */
                G.corpus_set_tie_order(hs[i], tie_order_for(&t->opt, i == 1 ? VG_TYPE_U8 : t->opt.v_type));
                G.corpus_set_scan_filter(hs[i], t->opt.scan_filter);
            }
        }
        return;
    }
    if (vc->count >= MAX_TABLES) { ctx_error(ctx, SQLITE_ERROR, "Cannot add table: maximum number of allowed tables reached (%d).", MAX_TABLES); return; }
    int without_rowid = table_without_rowid(db, tbl);
    char *pk = without_rowid ? single_int_pk(db, tbl) : dup_str("rowid");
    if (!pk) {
        if (without_rowid) ctx_error(ctx, SQLITE_ERROR, "WITHOUT ROWID table '%s' must have exactly one PRIMARY KEY column of type INTEGER.", tbl);
        else ctx_error(ctx, SQLITE_NOMEM, "Out of memory: unable to duplicate rowid column name.");
        return;
    }
    t = &vc->tables[vc->count];
    memset(t, 0, sizeof(*t));
    t->t_name = dup_str(tbl);
    t->c_name = dup_str(col);
    t->pk_name = pk;
    if (!t->t_name || !t->c_name) {
        sqlite3_free(t->t_name); sqlite3_free(t->c_name); sqlite3_free(pk);
        ctx_error(ctx, SQLITE_NOMEM, "Out of memory: unable to duplicate table or column name.");
        return;
    }
    t->opt = o;
    t->hook_seen = -1;
    vc->count++;
    meta_load(db, t);
    if (track_wanted(&t->opt)) track_install(db, vc);
}

/* ------------------------------------------------------------------------------------------------ quantization */

static int flush_chunk(sqlite3 *db, table_ctx *t, uint32_t n, const uint8_t *data, int64_t bytes, int64_t first, int64_t last) {
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "INSERT INTO vector0_%q_%q (rowid1, rowid2, counter, data) VALUES (?, ?, ?, ?);", t->t_name, t->c_name);
    sqlite3_stmt *st = NULL;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, NULL);
    if (rc == SQLITE_OK) {
        sqlite3_bind_int64(st, 1, first);
        sqlite3_bind_int64(st, 2, last);
        sqlite3_bind_int(st, 3, (int)n);
        sqlite3_bind_blob(st, 4, data, (int)bytes, SQLITE_STATIC);
        rc = sqlite3_step(st);
        if (rc == SQLITE_DONE) rc = SQLITE_OK;
    }
    sqlite3_finalize(st);
    return rc;
}

/* vector_quantize with a GPU present: the raw vectors are staged into HBM once (the same corpus later serves
 * vector_full_scan), min/max and the quantization run as kernels over it (vg_shards_minmax /
 * vg_shards_quantize_rows, bit-exact with the host arithmetic below), and only the persisted records are assembled
 * here: [int64 LE rowid | dim bytes] per row, flushed in max_memory-sized chunks exactly like the reference
 * (sqlite-vector.c:1282-1327).  Returns -1 when the GPU path cannot be used (caller runs the host passes). */
static int rebuild_quantization_gpu(sqlite3_context *ctx, table_ctx *t, int qtype, uint64_t max_memory, uint32_t *count) {
    sqlite3 *db = sqlite3_context_db_handle(ctx);
    if (!gpu_load() || G.device_count() <= 0) return -1;
    char *err = NULL;
    int rc = stage_full(db, (vec_context *)sqlite3_user_data(ctx), t, &err);
    if (rc != SQLITE_OK) {
        ctx_error(ctx, rc, "%s", err ? err : "staging failed");
        sqlite3_free(err);
        return rc;
    }
    t->full_validated = 1;
    const int64_t n = G.corpus_rows(t->full);
    if (n <= 0) return -1;                                    /* empty table: the host path has the reference's defaults */
    const int dim = t->opt.v_dim;
    const int64_t rec = 8 + (int64_t)dim;
    float lo, hi;
    int negative;
    if (G.corpus_minmax(t->full, &lo, &hi, &negative) != VG_OK) { ctx_error(ctx, SQLITE_ERROR, "%s", gpu_error()); return SQLITE_ERROR; }
    if (qtype == VG_QUANT_AUTO) qtype = negative ? VG_QUANT_S8 : VG_QUANT_U8;
    {
        float amax = fmaxf(fabsf(lo), fabsf(hi));
        t->scale = (qtype == VG_QUANT_U8) ? (255.0f / (hi - lo)) : (127.0f / amax);
        t->offset = (qtype == VG_QUANT_U8) ? lo : 0.0f;
        t->opt.q_type = qtype;
    }
    if (max_memory == 0) max_memory = (uint64_t)n * (uint64_t)rec;
    int64_t per_chunk = (int64_t)(max_memory / (uint64_t)rec);
    if (per_chunk <= 0) per_chunk = 1;
    if (per_chunk > n) per_chunk = n;
    uint8_t *qbuf = (uint8_t *)sqlite3_malloc64((sqlite3_uint64)per_chunk * (sqlite3_uint64)dim);
    uint8_t *rbuf = (uint8_t *)sqlite3_malloc64((sqlite3_uint64)per_chunk * (sqlite3_uint64)rec);
    if (!qbuf || !rbuf) { sqlite3_free(qbuf); sqlite3_free(rbuf); return SQLITE_NOMEM; }
    *count = 0;
    for (int64_t r0 = 0; r0 < n && rc == SQLITE_OK; r0 += per_chunk) {
        const int64_t nr = (n - r0 < per_chunk) ? (n - r0) : per_chunk;
        if (G.corpus_quantize_rows(t->full, t->scale, t->offset, qtype, r0, nr, qbuf) != VG_OK) {
            ctx_error(ctx, SQLITE_ERROR, "%s", gpu_error());
            rc = SQLITE_ERROR;
            break;
        }
        int64_t first = 0, last = 0;
        for (int64_t i = 0; i < nr; ++i) {
            const int64_t id = G.corpus_rowid_at(t->full, r0 + i);
            uint8_t *w = rbuf + i * rec;
            for (int b = 0; b < 8; ++b) w[b] = (uint8_t)(((uint64_t)id >> (8 * b)) & 0xFF);
            memcpy(w + 8, qbuf + i * dim, (size_t)dim);
            if (i == 0) first = id;
            last = id;
        }
        rc = flush_chunk(db, t, (uint32_t)nr, rbuf, nr * rec, first, last);
        *count += (uint32_t)nr;
    }
    sqlite3_free(qbuf);
    sqlite3_free(rbuf);
    return rc;
}

/* Two passes over "SELECT pk, col FROM tbl ORDER BY pk" (sqlite-vector.c:1009): global min/max (+ any negative),
 * then quantize every row into [int64 LE rowid | dim bytes] records flushed in max_memory-sized chunks.
 * Same arithmetic and the same persisted bytes as the reference (:1147-1336); the per-element quantizer is the
 * C-ABI's host routine (bit-exact, tests/test_abi_exports.py). */
static int rebuild_quantization(sqlite3_context *ctx, table_ctx *t, int qtype, uint64_t max_memory, uint32_t *count) {
    sqlite3 *db = sqlite3_context_db_handle(ctx);
    const int dim = t->opt.v_dim, type = t->opt.v_type, es = elem_size(type);
    const int64_t rec = 8 + (int64_t)dim;
    char sql[SQL_BUF];
    *count = 0;
    if (!gpu_load()) { ctx_error(ctx, SQLITE_ERROR, "%s", gpu_error()); return SQLITE_ERROR; }
    {
        int grc = rebuild_quantization_gpu(ctx, t, qtype, max_memory, count);
        if (grc != -1) return grc;                           /* done (or failed) on the GPU; -1 = host passes below */
    }
    if (max_memory == 0) {
        sqlite3_snprintf(sizeof(sql), sql, "SELECT COUNT(*) FROM %q;", t->t_name);
        int64_t n = read_int64(db, sql);
        if (n <= 0) { t->opt.q_type = (qtype == VG_QUANT_AUTO) ? VG_QUANT_U8 : qtype; t->scale = 1.0f; t->offset = 0.0f; return SQLITE_OK; }
        max_memory = (uint64_t)n * (uint64_t)rec;
    }
    uint32_t per_chunk = (uint32_t)(max_memory / (uint64_t)rec);
    if (per_chunk == 0) per_chunk = 1;
    uint8_t *buf = (uint8_t *)sqlite3_malloc64((sqlite3_uint64)per_chunk * (sqlite3_uint64)rec);
    if (!buf) return SQLITE_NOMEM;
    sqlite3_snprintf(sizeof(sql), sql, "SELECT %q, %q FROM %q ORDER BY %q;", t->pk_name, t->c_name, t->t_name, t->pk_name);
    sqlite3_stmt *st = NULL;
    int rc = sqlite3_prepare_v2(db, sql, -1, &st, NULL);
    if (rc != SQLITE_OK) { sqlite3_free(buf); return rc; }

    float lo = FLT_MAX, hi = -FLT_MAX;
    int negative = 0;
    while ((rc = sqlite3_step(st)) == SQLITE_ROW) {
        if (sqlite3_column_type(st, 1) == SQLITE_NULL) continue;
        const void *blob = sqlite3_column_blob(st, 1);
        if (!blob) continue;
        if (sqlite3_column_bytes(st, 1) < dim * es) {
            ctx_error(ctx, SQLITE_ERROR, "Invalid vector blob found at rowid %lld.", (long long)sqlite3_column_int64(st, 0));
            rc = SQLITE_ERROR;
            goto done;
        }
        for (int i = 0; i < dim; ++i) {
            float v = elem_as_f32(type, blob, i);
            if (v < lo) lo = v;
            if (v > hi) hi = v;
            if (v < 0.0) negative = 1;
        }
    }
    if (rc != SQLITE_DONE) goto done;
    if (qtype == VG_QUANT_AUTO) qtype = negative ? VG_QUANT_S8 : VG_QUANT_U8;
    {
        float amax = fmaxf(fabsf(lo), fabsf(hi));
        t->scale = (qtype == VG_QUANT_U8) ? (255.0f / (hi - lo)) : (127.0f / amax);
        t->offset = (qtype == VG_QUANT_U8) ? lo : 0.0f;
        t->opt.q_type = qtype;
    }
    rc = sqlite3_reset(st);
    if (rc != SQLITE_OK) goto done;
    {
        uint32_t n = 0;
        int64_t first = 0, last = 0;
        uint8_t *w = buf;
        while ((rc = sqlite3_step(st)) == SQLITE_ROW) {
            if (sqlite3_column_type(st, 1) == SQLITE_NULL) continue;
            const void *blob = sqlite3_column_blob(st, 1);
            if (!blob) continue;
            int64_t id = sqlite3_column_int64(st, 0);
            if (n == 0) first = id;
            for (int b = 0; b < 8; ++b) w[b] = (uint8_t)(((uint64_t)id >> (8 * b)) & 0xFF);
            G.quantize_query(type, blob, dim, t->scale, t->offset, qtype, w + 8);
            w += rec;
            last = id;
            ++n;
            ++*count;
            if (n == per_chunk) {
                rc = flush_chunk(db, t, n, buf, w - buf, first, last);
                if (rc != SQLITE_OK) goto done;
                n = 0;
                w = buf;
            }
        }
        if (rc != SQLITE_DONE) goto done;
        rc = n ? flush_chunk(db, t, n, buf, w - buf, first, last) : SQLITE_OK;
    }
done:
    sqlite3_finalize(st);
    sqlite3_free(buf);
    return rc;
}

static void do_preload(sqlite3_context *ctx, const char *tbl, const char *col);

static void quantize_common(sqlite3_context *ctx, const char *tbl, const char *col, const char *opts) {
    vec_context *vc = (vec_context *)sqlite3_user_data(ctx);
    table_ctx *t = context_lookup(vc, tbl, col);
    if (!t) {
        ctx_error(ctx, SQLITE_ERROR, "Vector context not found for table '%s' and column '%s'. Ensure that vector_init() has been called before using vector_quantize().", tbl, col);
        return;
    }
    sqlite3 *db = sqlite3_context_db_handle(ctx);
    char sql[SQL_BUF];
    uint32_t counter = 0;
    int stamps_were_fresh = 0;
    t->full_validated = 0;
    last_ctx_error[0] = 0;
    int rc = sqlite3_exec(db, "BEGIN;", NULL, NULL, NULL);          /* like the reference: fails inside a transaction */
    if (rc == SQLITE_OK) {
        sqlite3_snprintf(sizeof(sql), sql, "DROP TABLE IF EXISTS vector0_%q_%q;", tbl, col);
        rc = sqlite3_exec(db, sql, NULL, NULL, NULL);
    }
    if (rc == SQLITE_OK) {
        sqlite3_snprintf(sizeof(sql), sql, "CREATE TABLE IF NOT EXISTS vector0_%q_%q (rowid1 INTEGER, rowid2 INTEGER, counter INTEGER, data BLOB);", tbl, col);
        rc = sqlite3_exec(db, sql, NULL, NULL, NULL);
    }
    if (rc == SQLITE_OK) {
        vec_options o = t->opt;
        if (!options_parse(ctx, opts, &o)) { sqlite3_exec(db, "ROLLBACK;", NULL, NULL, NULL); return; }
        rc = rebuild_quantization(ctx, t, o.q_type, o.max_memory, &counter);
        stamps_were_fresh = (rc == SQLITE_OK && t->full && t->full_validated);
    }
    if (rc == SQLITE_OK) rc = sqlite3_exec(db, "COMMIT;", NULL, NULL, NULL);
    if (rc == SQLITE_OK) rc = meta_put(ctx, tbl, col, "qtype", 1, t->opt.q_type, 0);
    if (rc == SQLITE_OK) rc = meta_put(ctx, tbl, col, "qscale", 0, 0, t->scale);
    if (rc == SQLITE_OK) rc = meta_put(ctx, tbl, col, "qoffset", 0, 0, t->offset);
    if (rc != SQLITE_OK) {
        char *msg = sqlite3_mprintf("%s", last_ctx_error[0] ? last_ctx_error : sqlite3_errmsg(db));
        sqlite3_exec(db, "ROLLBACK;", NULL, NULL, NULL);
        if (msg) { sqlite3_result_error(ctx, msg, -1); sqlite3_free(msg); }
        sqlite3_result_error_code(ctx, rc);
        return;
    }
    int was_preloaded = t->quant_preloaded;
    if (t->quant) { G.corpus_destroy(t->quant); t->quant = NULL; }   /* HBM copy is stale now */
    if (t->full && stamps_were_fresh) {       /* only our own shadow-table writes happened, and they are committed */
        db_stamps(db, &t->full_data_version, &t->full_changes, &t->full_schema);
        t->full_in_txn = 0;
        if (t->hook_seen >= 0) t->hook_seen = ((vec_context *)sqlite3_user_data(ctx))->hook_events;   /* (the shadow-table rows were reported too) */
    }
    sqlite3_result_int64(ctx, (sqlite3_int64)counter);
    if (was_preloaded) do_preload(ctx, tbl, col);
}

static void fn_quantize3(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    static const int types[] = {SQLITE_TEXT, SQLITE_TEXT, SQLITE_TEXT};
    if (!check_args(ctx, "vector_quantize", argc, argv, 3, types)) return;
    quantize_common(ctx, (const char *)sqlite3_value_text(argv[0]), (const char *)sqlite3_value_text(argv[1]), (const char *)sqlite3_value_text(argv[2]));
}

static void fn_quantize2(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    static const int types[] = {SQLITE_TEXT, SQLITE_TEXT};
    if (!check_args(ctx, "vector_quantize", argc, argv, 2, types)) return;
    quantize_common(ctx, (const char *)sqlite3_value_text(argv[0]), (const char *)sqlite3_value_text(argv[1]), NULL);
}

static void fn_quantize_memory(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    static const int types[] = {SQLITE_TEXT, SQLITE_TEXT};
    if (!check_args(ctx, "vector_quantize_memory", argc, argv, 2, types)) return;
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "SELECT SUM(LENGTH(data)) FROM vector0_%q_%q;", (const char *)sqlite3_value_text(argv[0]), (const char *)sqlite3_value_text(argv[1]));
    sqlite3_result_int64(ctx, read_int64(sqlite3_context_db_handle(ctx), sql));
}

/* vector_quantize_preload: the reference concatenates every chunk into one host buffer; here the chunks go to HBM */
static void do_preload(sqlite3_context *ctx, const char *tbl, const char *col) {
    vec_context *vc = (vec_context *)sqlite3_user_data(ctx);
    table_ctx *t = context_lookup(vc, tbl, col);
    if (!t) {
        ctx_error(ctx, SQLITE_ERROR, "Vector context not found for table '%s' and column '%s'. Ensure that vector_init() has been called before using vector_quantize_preload().", tbl, col);
        return;
    }
    sqlite3 *db = sqlite3_context_db_handle(ctx);
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "SELECT SUM(LENGTH(data)) FROM vector0_%q_%q;", tbl, col);
    if (read_int64(db, sql) == 0) {
        ctx_error(ctx, SQLITE_ERROR, "Unable to read data from database. Ensure that vector_quantize() has been called before using vector_quantize_preload().");
        return;
    }
    char *err = NULL;
    int rc = stage_quant(db, t, 1, &err);
    if (rc != SQLITE_OK) {
        ctx_error(ctx, rc, "vector_quantize_preload: %s", err ? err : "staging failed");
        sqlite3_free(err);
        return;
    }
    t->quant_preloaded = 1;
}

static void fn_quantize_preload(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    static const int types[] = {SQLITE_TEXT, SQLITE_TEXT};
    if (!check_args(ctx, "vector_quantize_preload", argc, argv, 2, types)) return;
    do_preload(ctx, (const char *)sqlite3_value_text(argv[0]), (const char *)sqlite3_value_text(argv[1]));
}

static void fn_quantize_cleanup(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    static const int types[] = {SQLITE_TEXT, SQLITE_TEXT};
    if (!check_args(ctx, "vector_quantize_cleanup", argc, argv, 2, types)) return;
    const char *tbl = (const char *)sqlite3_value_text(argv[0]);
    const char *col = (const char *)sqlite3_value_text(argv[1]);
    table_ctx *t = context_lookup((vec_context *)sqlite3_user_data(ctx), tbl, col);
    if (!t) return;
    if (t->quant) { G.corpus_destroy(t->quant); t->quant = NULL; }
    t->quant_preloaded = 0;
    char sql[SQL_BUF];
    sqlite3_snprintf(sizeof(sql), sql, "DROP TABLE IF EXISTS vector0_%q_%q;", tbl, col);
    sqlite3_exec(sqlite3_context_db_handle(ctx), sql, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------------------------------------ table-valued functions */

typedef struct {
    sqlite3_vtab base;
    sqlite3 *db;
    vec_context *ctx;
} scan_vtab;

typedef struct {
    sqlite3_vtab_cursor base;
    int streaming;
    /* materialised top-k */
    int64_t *rowids;
    double *distance;
    int row_index, row_count;
    int *query_no;                     /* batch TVFs: which query of the batch each output row answers */
    /* streaming: all N distances computed by ONE kernel launch, paged out row by row */
    float *all_dist;
    int64_t *all_rowids;               /* the cursor's own snapshot: the corpus may be re-staged / destroyed while it is stepped */
    int64_t stream_pos, stream_n;
} scan_cursor;

static int tvf_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    int rc = sqlite3_declare_vtab(db, "CREATE TABLE x(tbl hidden, vector hidden, k hidden, memidx hidden, id, distance);");
    if (rc != SQLITE_OK) return rc;
    scan_vtab *v = (scan_vtab *)sqlite3_malloc(sizeof(scan_vtab));
    if (!v) return SQLITE_NOMEM;
    memset(v, 0, sizeof(*v));
    v->db = db;
    v->ctx = (vec_context *)aux;
    *out = &v->base;
    return SQLITE_OK;
}

static int tvf_disconnect(sqlite3_vtab *v) { sqlite3_free(v); return SQLITE_OK; }

static void map_constraints(sqlite3_index_info *info) {
    for (int i = 0; i < info->nConstraint; ++i) {
        const struct sqlite3_index_constraint *c = &info->aConstraint[i];
        if (!c->usable || c->op != SQLITE_INDEX_CONSTRAINT_EQ) continue;
        if (c->iColumn >= COL_TBL && c->iColumn <= COL_MEMIDX) {
            info->aConstraintUsage[i].argvIndex = c->iColumn + 1;
            info->aConstraintUsage[i].omit = 1;
        }
    }
}

static int topk_best_index(sqlite3_vtab *v, sqlite3_index_info *info) {
    info->estimatedCost = 1.0;
    info->estimatedRows = 100;
    info->orderByConsumed = 1;                   /* output is distance-ascending (sqlite-vector.c:1853) */
    info->idxNum = 1;
    map_constraints(info);
    return SQLITE_OK;
}

static int stream_best_index(sqlite3_vtab *v, sqlite3_index_info *info) {
    info->estimatedCost = 1e8;                   /* no ordering promise (sqlite-vector.c:2245-2275) */
    info->estimatedRows = 100000;
    map_constraints(info);
    return SQLITE_OK;
}

static int tvf_open(sqlite3_vtab *v, sqlite3_vtab_cursor **out) {
    scan_cursor *c = (scan_cursor *)sqlite3_malloc(sizeof(scan_cursor));
    if (!c) return SQLITE_NOMEM;
    memset(c, 0, sizeof(*c));
    *out = &c->base;
    return SQLITE_OK;
}

static int tvf_close(sqlite3_vtab_cursor *cur) {
    scan_cursor *c = (scan_cursor *)cur;
    sqlite3_free(c->rowids);
    sqlite3_free(c->distance);
    sqlite3_free(c->all_dist);
    sqlite3_free(c->all_rowids);
    sqlite3_free(c->query_no);
    sqlite3_free(c);
    return SQLITE_OK;
}

/* the common xFilter (reference: vCursorFilterCommon, sqlite-vector.c:1723-1826) */
static int filter_common(sqlite3_vtab_cursor *cur, int argc, sqlite3_value **argv, const char *fname, int streaming, int quantized) {
    scan_cursor *c = (scan_cursor *)cur;
    scan_vtab *vt = (scan_vtab *)cur->pVtab;
    c->streaming = streaming;
    c->row_index = 0;
    c->row_count = 0;
    c->stream_pos = 0;
    c->stream_n = 0;
    const int nargs = streaming ? 3 : 4;
    if (argc != nargs) return vtab_error(&vt->base, "%s expects %d arguments, but %d were provided.", fname, nargs, argc);
    for (int i = 0; i < argc; ++i) {
        int t = sqlite3_value_type(argv[i]);
        if (i < 2 && t != SQLITE_TEXT) return vtab_error(&vt->base, "%s: argument %d must be of type TEXT (got %s).", fname, i + 1, sql_type_name(t));
        if (i == 2 && t != SQLITE_TEXT && t != SQLITE_BLOB) return vtab_error(&vt->base, "%s: argument %d must be of type TEXT or BLOB (got %s).", fname, i + 1, sql_type_name(t));
        if (i == 3 && t != SQLITE_INTEGER) return vtab_error(&vt->base, "%s: argument %d must be of type INTEGER (got %s).", fname, i + 1, sql_type_name(t));
    }
    const char *tbl = (const char *)sqlite3_value_text(argv[0]);
    const char *col = (const char *)sqlite3_value_text(argv[1]);
    table_ctx *t = context_lookup(vt->ctx, tbl, col);
    if (!t) return vtab_error(&vt->base, "%s: unable to retrieve context.", fname);

    const void *query = NULL;
    void *owned = NULL;
    int qbytes = 0;
    if (sqlite3_value_type(argv[2]) == SQLITE_TEXT) {
        owned = vector_from_json(NULL, &vt->base, t->opt.v_type, (const char *)sqlite3_value_text(argv[2]), &qbytes, t->opt.v_dim);
        if (!owned) return SQLITE_ERROR;
        query = owned;
    } else {
        query = sqlite3_value_blob(argv[2]);
        qbytes = sqlite3_value_bytes(argv[2]);
        if (!query) return vtab_error(&vt->base, "%s: input vector cannot be NULL.", fname);
    }
    int rc = SQLITE_OK;
    char *err = NULL;
    uint8_t *qquant = NULL;
    if (qbytes < t->opt.v_dim * elem_size(t->opt.v_type)) {
        /* the reference reads past a short query BLOB (:1774); refuse instead of faulting on the device */
        rc = vtab_error(&vt->base, "%s: query vector has %d bytes, expected %d.", fname, qbytes, t->opt.v_dim * elem_size(t->opt.v_type));
        goto out;
    }
    if (quantized) {
        char name[SQL_BUF];
        sqlite3_snprintf(sizeof(name), name, "vector0_%q_%q", tbl, col);
        if (!exists_in_master(vt->db, "table", name)) {
            rc = vtab_error(&vt->base, "Quantization table not found for table '%s' and column '%s'. Ensure that vector_quantize() has been called before using vector_quantize_scan().", tbl, col);
            goto out;
        }
    }
    int k = streaming ? 0 : sqlite3_value_int(argv[3]);
    if (!streaming && k == 0) { rc = SQLITE_DONE; goto out; }       /* sqlite-vector.c:1796 (ends the statement) */
    if (!streaming && k < 0) { rc = vtab_error(&vt->base, "%s: k must be positive.", fname); goto out; }

    /* the corpus to scan + the query in its element type */
    vg_shards *corpus = NULL;
    const void *scan_query = query;
    if (quantized) {
        if (!t->quant_preloaded || !t->quant) rc = stage_quant(vt->db, t, 0, &err);   /* not preloaded: stage on first use */
        if (rc != SQLITE_OK) { rc = vtab_error(&vt->base, "%s: %s", fname, err ? err : "staging failed"); goto out; }
        qquant = (uint8_t *)sqlite3_malloc(t->opt.v_dim);
        if (!qquant) { rc = SQLITE_NOMEM; goto out; }
        /* the query is quantized with the stored scale/offset on every scan (sqlite-vector.c:2166-2177) */
        if (G.quantize_query(t->opt.v_type, query, t->opt.v_dim, t->scale, t->offset, t->opt.q_type, qquant) != VG_OK) {
            rc = vtab_error(&vt->base, "%s: %s", fname, gpu_error());
            goto out;
        }
        scan_query = qquant;
        corpus = t->quant;
    } else {
        rc = stage_full(vt->db, vt->ctx, t, &err);
        if (rc != SQLITE_OK) { rc = vtab_error(&vt->base, "%s: %s", fname, err ? err : "staging failed"); goto out; }
        corpus = t->full;
    }

    if (streaming) {
        int64_t n = G.corpus_rows(corpus);
        sqlite3_free(c->all_dist);
        sqlite3_free(c->all_rowids);
        c->all_dist = (float *)sqlite3_malloc64((sqlite3_uint64)(n > 0 ? n : 1) * sizeof(float));
        c->all_rowids = (int64_t *)sqlite3_malloc64((sqlite3_uint64)(n > 0 ? n : 1) * sizeof(int64_t));
        if (!c->all_dist || !c->all_rowids) { rc = SQLITE_NOMEM; goto out; }
        /* distances AND rowids are copied into the cursor here (the reference's stream cursor reads its own statement,
         * sqlite-vector.c:2277-2313): later re-staging / vector_quantize / cleanup on the table cannot touch them */
        if (n > 0 && (G.scan_distances(corpus, t->opt.v_distance, scan_query, c->all_dist) != VG_OK ||
                      G.corpus_rowids(corpus, 0, n, c->all_rowids) != VG_OK)) {
            rc = vtab_error(&vt->base, "%s: %s", fname, gpu_error());
            goto out;
        }
        c->stream_n = n;
        c->stream_pos = 0;
    } else {
        sqlite3_free(c->rowids);
        sqlite3_free(c->distance);
        c->rowids = (int64_t *)sqlite3_malloc64((sqlite3_uint64)k * sizeof(int64_t));
        c->distance = (double *)sqlite3_malloc64((sqlite3_uint64)k * sizeof(double));
        if (!c->rowids || !c->distance) { rc = SQLITE_NOMEM; goto out; }
        int got = 0;
        if (G.scan_topk(corpus, t->opt.v_distance, scan_query, k, c->rowids, c->distance, &got) != VG_OK) {
            rc = vtab_error(&vt->base, "%s: %s", fname, gpu_error());
            goto out;
        }
        c->row_count = got;                       /* fewer than k when fewer rows qualify (:1816-1817) */
    }
out:
    sqlite3_free(err);
    sqlite3_free(owned);
    sqlite3_free(qquant);
    return rc;
}

static int full_filter(sqlite3_vtab_cursor *c, int n, const char *s, int argc, sqlite3_value **argv) { return filter_common(c, argc, argv, "vector_full_scan", 0, 0); }
static int quant_filter(sqlite3_vtab_cursor *c, int n, const char *s, int argc, sqlite3_value **argv) { return filter_common(c, argc, argv, "vector_quantize_scan", 0, 1); }
static int full_stream_filter(sqlite3_vtab_cursor *c, int n, const char *s, int argc, sqlite3_value **argv) { return filter_common(c, argc, argv, "vector_full_scan_stream", 1, 0); }
static int quant_stream_filter(sqlite3_vtab_cursor *c, int n, const char *s, int argc, sqlite3_value **argv) { return filter_common(c, argc, argv, "vector_quantize_scan_stream", 1, 1); }


/* ---- batched queries (the reference has no multi-query entry point: its equivalent is nq separate
 * vector_full_scan statements).  vector_full_scan_batch(tbl, col, queries, k) / vector_quantize_scan_batch(...):
 *   queries  BLOB of nq * dim elements (query vectors back to back), or TEXT JSON array of arrays
 *   output   (query, id, distance): k rows per query, ordered by query number (0-based) then distance ascending
 * Each query's rows are what the single-query function returns for it; f32 DOT/COSINE batches run as one
 * matrix-core pass over the corpus (vg_scan_topk_batch), everything else as nq GPU scans. */

static int batch_connect(sqlite3 *db, void *aux, int argc, const char *const *argv, sqlite3_vtab **out, char **err) {
    int rc = sqlite3_declare_vtab(db, "CREATE TABLE x(tbl hidden, vector hidden, k hidden, memidx hidden, id, distance, query);");
    if (rc != SQLITE_OK) return rc;
    scan_vtab *v = (scan_vtab *)sqlite3_malloc(sizeof(scan_vtab));
    if (!v) return SQLITE_NOMEM;
    memset(v, 0, sizeof(*v));
    v->db = db;
    v->ctx = (vec_context *)aux;
    *out = &v->base;
    return SQLITE_OK;
}

static int batch_best_index(sqlite3_vtab *v, sqlite3_index_info *info) {
    info->estimatedCost = 10.0;
    info->estimatedRows = 1000;
    info->idxNum = 2;
    map_constraints(info);
    /* rows come out as (query asc, distance asc): claim the order only when that is what was asked for */
    if (info->nOrderBy == 2 && info->aOrderBy[0].iColumn == COL_QUERY && !info->aOrderBy[0].desc &&
        info->aOrderBy[1].iColumn == COL_DISTANCE && !info->aOrderBy[1].desc) info->orderByConsumed = 1;
    if (info->nOrderBy == 1 && info->aOrderBy[0].iColumn == COL_QUERY && !info->aOrderBy[0].desc) info->orderByConsumed = 1;
    return SQLITE_OK;
}

/* "[[..],[..]]" -> nq vectors back to back; returns sqlite3_malloc'd buffer */
static void *batch_from_json(sqlite3_vtab *vt, int type, const char *json, int dim, int *nq_out) {
    const int es = elem_size(type);
    const char *p = json;
    while (*p && isspace((unsigned char)*p)) p++;
    if (*p != '[') { vtab_error(vt, "Malformed JSON: expected '[' at the beginning of the array."); return NULL; }
    p++;
    int cap = 0, nq = 0;
    char *buf = NULL;
    while (1) {
        while (*p && (isspace((unsigned char)*p) || *p == ',')) p++;
        if (*p == ']' || !*p) break;
        if (*p != '[') { vtab_error(vt, "Malformed JSON: expected an array of arrays."); sqlite3_free(buf); return NULL; }
        int sz = 0;
        void *one = vector_from_json(NULL, vt, type, p, &sz, dim);
        if (!one) { sqlite3_free(buf); return NULL; }
        if (nq == cap) {
            cap = cap ? cap * 2 : 16;
            char *nb = (char *)sqlite3_realloc64(buf, (sqlite3_uint64)cap * dim * es);
            if (!nb) { sqlite3_free(one); sqlite3_free(buf); return NULL; }
            buf = nb;
        }
        memcpy(buf + (size_t)nq * dim * es, one, (size_t)dim * es);
        sqlite3_free(one);
        nq++;
        while (*p && *p != ']') p++;
        if (*p == ']') p++;
    }
    *nq_out = nq;
    return buf;
}

static int batch_filter_common(sqlite3_vtab_cursor *cur, int argc, sqlite3_value **argv, const char *fname, int quantized) {
    scan_cursor *c = (scan_cursor *)cur;
    scan_vtab *vt = (scan_vtab *)cur->pVtab;
    c->streaming = 0;
    c->row_index = 0;
    c->row_count = 0;
    if (argc != 4) return vtab_error(&vt->base, "%s expects %d arguments, but %d were provided.", fname, 4, argc);
    for (int i = 0; i < argc; ++i) {
        int t = sqlite3_value_type(argv[i]);
        if (i < 2 && t != SQLITE_TEXT) return vtab_error(&vt->base, "%s: argument %d must be of type TEXT (got %s).", fname, i + 1, sql_type_name(t));
        if (i == 2 && t != SQLITE_TEXT && t != SQLITE_BLOB) return vtab_error(&vt->base, "%s: argument %d must be of type TEXT or BLOB (got %s).", fname, i + 1, sql_type_name(t));
        if (i == 3 && t != SQLITE_INTEGER) return vtab_error(&vt->base, "%s: argument %d must be of type INTEGER (got %s).", fname, i + 1, sql_type_name(t));
    }
    const char *tbl = (const char *)sqlite3_value_text(argv[0]);
    const char *col = (const char *)sqlite3_value_text(argv[1]);
    table_ctx *t = context_lookup(vt->ctx, tbl, col);
    if (!t) return vtab_error(&vt->base, "%s: unable to retrieve context.", fname);
    const int dim = t->opt.v_dim, es = elem_size(t->opt.v_type);
    const int64_t qrow = (int64_t)dim * es;

    const uint8_t *queries = NULL;
    void *owned = NULL;
    uint8_t *qquant = NULL;
    int64_t *ids = NULL;
    double *dist = NULL;
    int *counts = NULL;
    char *err = NULL;
    int nq = 0, rc = SQLITE_OK;
    if (sqlite3_value_type(argv[2]) == SQLITE_TEXT) {
        owned = batch_from_json(&vt->base, t->opt.v_type, (const char *)sqlite3_value_text(argv[2]), dim, &nq);
        if (!owned && nq == 0 && vt->base.zErrMsg) return SQLITE_ERROR;
        queries = (const uint8_t *)owned;
    } else {
        queries = (const uint8_t *)sqlite3_value_blob(argv[2]);
        const int64_t bytes = sqlite3_value_bytes(argv[2]);
        if (!queries || bytes == 0 || bytes % qrow != 0)
            return vtab_error(&vt->base, "%s: the query batch has %lld bytes, expected a multiple of %lld (dimension %d).", fname, (long long)bytes, (long long)qrow, dim);
        nq = (int)(bytes / qrow);
    }
    const int k = sqlite3_value_int(argv[3]);
    if (k == 0 || nq == 0) { rc = (k == 0) ? SQLITE_DONE : SQLITE_OK; goto out; }
    if (k < 0) { rc = vtab_error(&vt->base, "%s: k must be positive.", fname); goto out; }

    vg_shards *corpus = NULL;
    const void *scan_queries = queries;
    if (quantized) {
        char name[SQL_BUF];
        sqlite3_snprintf(sizeof(name), name, "vector0_%q_%q", tbl, col);
        if (!exists_in_master(vt->db, "table", name)) {
            rc = vtab_error(&vt->base, "Quantization table not found for table '%s' and column '%s'. Ensure that vector_quantize() has been called before using vector_quantize_scan().", tbl, col);
            goto out;
        }
        if (!t->quant_preloaded || !t->quant) rc = stage_quant(vt->db, t, 0, &err);
        if (rc != SQLITE_OK) { rc = vtab_error(&vt->base, "%s: %s", fname, err ? err : "staging failed"); goto out; }
        qquant = (uint8_t *)sqlite3_malloc64((sqlite3_uint64)nq * dim);
        if (!qquant) { rc = SQLITE_NOMEM; goto out; }
        for (int i = 0; i < nq; ++i) {
            if (G.quantize_query(t->opt.v_type, queries + (int64_t)i * qrow, dim, t->scale, t->offset, t->opt.q_type, qquant + (int64_t)i * dim) != VG_OK) {
                rc = vtab_error(&vt->base, "%s: %s", fname, gpu_error());
                goto out;
            }
        }
        scan_queries = qquant;
        corpus = t->quant;
    } else {
        rc = stage_full(vt->db, vt->ctx, t, &err);
        if (rc != SQLITE_OK) { rc = vtab_error(&vt->base, "%s: %s", fname, err ? err : "staging failed"); goto out; }
        corpus = t->full;
    }
    {
        const int64_t n = G.corpus_rows(corpus);
        const int kk = (int)((int64_t)k < n ? (int64_t)k : (n > 0 ? n : 1));     /* never more than N rows per query */
        ids = (int64_t *)sqlite3_malloc64((sqlite3_uint64)nq * kk * sizeof(int64_t));
        dist = (double *)sqlite3_malloc64((sqlite3_uint64)nq * kk * sizeof(double));
        counts = (int *)sqlite3_malloc64((sqlite3_uint64)nq * sizeof(int));
        if (!ids || !dist || !counts) { rc = SQLITE_NOMEM; goto out; }
        if (G.scan_topk_batch(corpus, t->opt.v_distance, scan_queries, nq, kk, ids, dist, counts) != VG_OK) {
            rc = vtab_error(&vt->base, "%s: %s", fname, gpu_error());
            goto out;
        }
        int64_t total = 0;
        for (int i = 0; i < nq; ++i) total += counts[i];
        sqlite3_free(c->rowids); sqlite3_free(c->distance); sqlite3_free(c->query_no);
        c->rowids = (int64_t *)sqlite3_malloc64((sqlite3_uint64)(total ? total : 1) * sizeof(int64_t));
        c->distance = (double *)sqlite3_malloc64((sqlite3_uint64)(total ? total : 1) * sizeof(double));
        c->query_no = (int *)sqlite3_malloc64((sqlite3_uint64)(total ? total : 1) * sizeof(int));
        if (!c->rowids || !c->distance || !c->query_no) { rc = SQLITE_NOMEM; goto out; }
        int w = 0;
        for (int i = 0; i < nq; ++i)
            for (int j = 0; j < counts[i]; ++j, ++w) {
                c->rowids[w] = ids[(int64_t)i * kk + j];
                c->distance[w] = dist[(int64_t)i * kk + j];
                c->query_no[w] = i;
            }
        c->row_count = w;
    }
out:
    sqlite3_free(err);
    sqlite3_free(owned);
    sqlite3_free(qquant);
    sqlite3_free(ids);
    sqlite3_free(dist);
    sqlite3_free(counts);
    return rc;
}

static int full_batch_filter(sqlite3_vtab_cursor *c, int n, const char *s, int argc, sqlite3_value **argv) { return batch_filter_common(c, argc, argv, "vector_full_scan_batch", 0); }
static int quant_batch_filter(sqlite3_vtab_cursor *c, int n, const char *s, int argc, sqlite3_value **argv) { return batch_filter_common(c, argc, argv, "vector_quantize_scan_batch", 1); }

static int tvf_next(sqlite3_vtab_cursor *cur) {
    scan_cursor *c = (scan_cursor *)cur;
    if (c->streaming) c->stream_pos++; else c->row_index++;
    return SQLITE_OK;
}

static int tvf_eof(sqlite3_vtab_cursor *cur) {
    scan_cursor *c = (scan_cursor *)cur;
    return c->streaming ? (c->stream_pos >= c->stream_n) : (c->row_index >= c->row_count);
}

static sqlite3_int64 cursor_rowid(scan_cursor *c) {
    return c->streaming ? (sqlite3_int64)c->all_rowids[c->stream_pos] : (sqlite3_int64)c->rowids[c->row_index];
}

static int tvf_column(sqlite3_vtab_cursor *cur, sqlite3_context *ctx, int col) {
    scan_cursor *c = (scan_cursor *)cur;
    if (col == COL_ID) sqlite3_result_int64(ctx, cursor_rowid(c));
    else if (col == COL_DISTANCE) sqlite3_result_double(ctx, c->streaming ? (double)c->all_dist[c->stream_pos] : c->distance[c->row_index]);
    else if (col == COL_QUERY && c->query_no) sqlite3_result_int(ctx, c->query_no[c->row_index]);
    return SQLITE_OK;
}

static int tvf_rowid(sqlite3_vtab_cursor *cur, sqlite3_int64 *out) {
    *out = cursor_rowid((scan_cursor *)cur);
    return SQLITE_OK;
}

#define SCAN_MODULE(name, best_index, filter)                                                        \
    static sqlite3_module name = {0, 0, tvf_connect, best_index, tvf_disconnect, 0, tvf_open, tvf_close, filter, \
                                  tvf_next, tvf_eof, tvf_column, tvf_rowid, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}

SCAN_MODULE(full_scan_module, topk_best_index, full_filter);
SCAN_MODULE(quant_scan_module, topk_best_index, quant_filter);
SCAN_MODULE(full_stream_module, stream_best_index, full_stream_filter);
SCAN_MODULE(quant_stream_module, stream_best_index, quant_stream_filter);
static sqlite3_module full_batch_module = {0, 0, batch_connect, batch_best_index, tvf_disconnect, 0, tvf_open, tvf_close, full_batch_filter,
                                           tvf_next, tvf_eof, tvf_column, tvf_rowid, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static sqlite3_module quant_batch_module = {0, 0, batch_connect, batch_best_index, tvf_disconnect, 0, tvf_open, tvf_close, quant_batch_filter,
                                            tvf_next, tvf_eof, tvf_column, tvf_rowid, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

/* ------------------------------------------------------------------------------------------------ registration */

static void fn_version(sqlite3_context *ctx, int argc, sqlite3_value **argv) { sqlite3_result_text(ctx, VECTOR_EXT_VERSION, -1, SQLITE_STATIC); }

static void fn_backend(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    if (gpu_load()) sqlite3_result_text(ctx, G.backend_name(), -1, SQLITE_TRANSIENT);
    else sqlite3_result_text(ctx, "HIP (engine not loaded)", -1, SQLITE_STATIC);
}

#ifdef _WIN32
__declspec(dllexport)
#endif
int sqlite3_vector_init(sqlite3 *db, char **pzErrMsg, const sqlite3_api_routines *pApi) {
    SQLITE_EXTENSION_INIT2(pApi);
    gpu_load();                                   /* best effort now; scans report the reason if it failed */
    int rc = sqlite3_exec(db, "CREATE TABLE IF NOT EXISTS _sqliteai_vector (tblname TEXT, colname TEXT, key TEXT, value ANY, PRIMARY KEY(tblname, colname, key));", NULL, NULL, NULL);
    if (rc != SQLITE_OK) return rc;
    vec_context *ctx = (vec_context *)sqlite3_malloc(sizeof(vec_context));
    if (!ctx) {
        if (pzErrMsg) *pzErrMsg = sqlite3_mprintf("Out of memory: failed to allocate vector extension context.");
        return SQLITE_NOMEM;
    }
    memset(ctx, 0, sizeof(*ctx));
    /* the context dies with the connection: vector_version owns it (same trick as sqlite-vector.c:2574) */
    rc = sqlite3_create_function_v2(db, "vector_version", 0, SQLITE_UTF8, ctx, fn_version, NULL, NULL, context_free);
    if (rc != SQLITE_OK) return rc;
    static const struct { const char *name; int nargs; void (*fn)(sqlite3_context *, int, sqlite3_value **); } fns[] = {
        {"vector_backend", 0, fn_backend},
        {"vector_init", 3, fn_vector_init},
        {"vector_quantize", 3, fn_quantize3},
        {"vector_quantize", 2, fn_quantize2},
        {"vector_quantize_memory", 2, fn_quantize_memory},
        {"vector_quantize_preload", 2, fn_quantize_preload},
        {"vector_quantize_cleanup", 2, fn_quantize_cleanup},
        {"vector_as_f32", 1, fn_as_f32}, {"vector_as_f32", 2, fn_as_f32},
        {"vector_as_f16", 1, fn_as_f16}, {"vector_as_f16", 2, fn_as_f16},
        {"vector_as_bf16", 1, fn_as_bf16}, {"vector_as_bf16", 2, fn_as_bf16},
        {"vector_as_i8", 1, fn_as_i8}, {"vector_as_i8", 2, fn_as_i8},
        {"vector_as_u8", 1, fn_as_u8}, {"vector_as_u8", 2, fn_as_u8},
    };
    for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); ++i) {
        rc = sqlite3_create_function(db, fns[i].name, fns[i].nargs, SQLITE_UTF8, ctx, fns[i].fn, NULL, NULL);
        if (rc != SQLITE_OK) return rc;
    }
    rc = sqlite3_create_module(db, "vector_full_scan", &full_scan_module, ctx);
    if (rc == SQLITE_OK) rc = sqlite3_create_module(db, "vector_quantize_scan", &quant_scan_module, ctx);
    if (rc == SQLITE_OK) rc = sqlite3_create_module(db, "vector_full_scan_stream", &full_stream_module, ctx);
    if (rc == SQLITE_OK) rc = sqlite3_create_module(db, "vector_quantize_scan_stream", &quant_stream_module, ctx);
    /* additions over the reference's surface (batched queries) */
    if (rc == SQLITE_OK) rc = sqlite3_create_module(db, "vector_full_scan_batch", &full_batch_module, ctx);
    if (rc == SQLITE_OK) rc = sqlite3_create_module(db, "vector_quantize_scan_batch", &quant_batch_module, ctx);
    return rc;
}
