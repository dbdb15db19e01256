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
 *
 * Layout: this file holds the includes, the registration and the entry point; the rest lives in vext_*.inc by concern
 * (gpulib, context, tracking, sqlutil, convert, staging, quantize, tvf, batch, cursor).  The JSON and option-string parsers
 * (vext_convert.inc / vext_sqlutil.inc) implement the reference's user-visible contract - what is accepted, every error
 * text - in this repository's own structure; a differential test against the reference extension pins that contract.
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
#include <time.h>
#include <unistd.h>
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

/* The extension is ONE translation unit (every function static, one exported symbol) kept in parts by concern: */
#include "vext_gpulib.inc"
#include "vext_shared.inc"
#include "vext_context.inc"
#include "vext_tracking.inc"
#include "vext_sqlutil.inc"
#include "vext_convert.inc"
#include "vext_staging.inc"
#include "vext_quantize.inc"
#include "vext_tvf.inc"
#include "vext_batch.inc"
#include "vext_cursor.inc"

/* ------------------------------------------------------------------------------------------------ registration */

static void fn_version(sqlite3_context *ctx, int argc, sqlite3_value **argv) { sqlite3_result_text(ctx, VECTOR_EXT_VERSION, -1, SQLITE_STATIC); }

static void fn_backend(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    if (gpu_load()) sqlite3_result_text(ctx, G.backend_name(), -1, SQLITE_TRANSIENT);
    else sqlite3_result_text(ctx, "HIP (engine not loaded)", -1, SQLITE_STATIC);
}

/* vector_gpu_memory(table, column): an addition over the reference's surface - what the table's copies hold on the device(s), as JSON text.
 * vector_quantize_memory() keeps the reference's meaning (the bytes of the persisted quantization); the shadow copies the filter scans make
 * (+ 26 % / + 52 % of an f32 / f16 corpus above 2^20 rows) and the batch kernels' tile-major copies are only visible here. */
static void fn_gpu_memory(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    static const int types[] = {SQLITE_TEXT, SQLITE_TEXT};
    if (!check_args(ctx, "vector_gpu_memory", argc, argv, 2, types)) return;
    const char *tbl = (const char *)sqlite3_value_text(argv[0]), *col = (const char *)sqlite3_value_text(argv[1]);
    vec_context *vc = (vec_context *)sqlite3_user_data(ctx);
    table_ctx *t = context_lookup(vc, tbl, col);
    if (!t) { ctx_error(ctx, SQLITE_ERROR, "Vector context not found for table '%s' and column '%s'. Ensure that vector_init() has been called before using vector_gpu_memory().", tbl, col); return; }
    long long f[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    if ((t->full || t->quant) && !gpu_load()) { ctx_error(ctx, SQLITE_ERROR, "%s", gpu_error()); return; }
    int frc = VG_OK, qrc = VG_OK, fsh = 0, qsh = 0;
    full_lock(t);
    if (t->full) frc = G.corpus_device_bytes(t->full, f);
    full_unlock(t);
    quant_lock(t);
    if (t->quant) qrc = G.corpus_device_bytes(t->quant, q);
    quant_unlock(t);
    if (frc != VG_OK || qrc != VG_OK) { ctx_error(ctx, SQLITE_ERROR, "%s", gpu_error()); return; }
    /* sharers: connections of this process holding this very copy (vext_shared.inc; 0 = the connection's own) - the bytes are held ONCE */
    pthread_mutex_lock(&g_shared_mu);
    if (t->full_sh) fsh = t->full_sh->refs;
    if (t->quant_sh) qsh = t->quant_sh->refs;
    pthread_mutex_unlock(&g_shared_mu);
    /* out_of_core: the table (its quantized records) did not fit the device at its last scan - nothing is resident, every scan reads the
     * rows again through two slabs of slab_rows rows (vext_staging.inc: ooc_plan) */
    char *js = sqlite3_mprintf("{\"column\":{\"staged\":%d,\"rows_bytes\":%lld,\"derived_bytes\":%lld,\"working_bytes\":%lld,\"out_of_core\":%d,\"slab_rows\":%lld,\"host_resident_bytes\":%lld,\"sharers\":%d},"
                               "\"quantized\":{\"staged\":%d,\"rows_bytes\":%lld,\"derived_bytes\":%lld,\"working_bytes\":%lld,\"out_of_core\":%d,\"slab_rows\":%lld,\"sharers\":%d},\"total_bytes\":%lld}",
                               t->full ? 1 : 0, f[0], f[1], f[2], t->full_ooc, (long long)(t->full_ooc ? t->full_slab_rows : 0),
                               (long long)(t->host_rows ? t->host_n * ((long long)elem_size(t->opt.v_type) * t->opt.v_dim + 8) : 0), fsh,
                               t->quant ? 1 : 0, q[0], q[1], q[2], t->quant_ooc, (long long)(t->quant_ooc ? t->quant_slab_rows : 0), qsh,
                               f[0] + f[1] + f[2] + q[0] + q[1] + q[2]);
    if (!js) { sqlite3_result_error_nomem(ctx); return; }
    sqlite3_result_text(ctx, js, -1, sqlite3_free);
}

/* vector_gpu_stats(): an addition over the reference's surface - what staging into HBM has cost this process, as JSON text */
static void fn_gpu_stats(sqlite3_context *ctx, int argc, sqlite3_value **argv) {
    const stage_stats g = stage_stats_read();
    pthread_mutex_lock(&g_stage_mu);
    const long long os = ooc_stat_scans, orows = ooc_stat_rows, hts = g_host_tier_scans, htf = g_host_tier_fills;
    pthread_mutex_unlock(&g_stage_mu);
    long long sh_n = 0, sh_refs = 0, sh_att, sh_pub;
    pthread_mutex_lock(&g_shared_mu);
    for (shared_corpus *e = g_shared; e; e = e->next) { ++sh_n; sh_refs += e->refs; }
    sh_att = g_shared_attached; sh_pub = g_shared_published;
    pthread_mutex_unlock(&g_shared_mu);
    char *js = sqlite3_mprintf("{\"stage_passes\":%lld,\"parallel_reader_passes\":%lld,\"rows_staged\":%lld,\"seconds_staging\":%.6f,\"seconds_in_engine_append\":%.6f,\"seconds_count_star\":%.6f,\"seconds_hbm_reserve\":%.6f,\"out_of_core_scans\":%lld,\"out_of_core_rows\":%lld,"
                               "\"host_tier_scans\":%lld,\"host_tier_fills\":%lld,"
                               "\"shared_copies\":%lld,\"shared_references\":%lld,\"shared_attachments\":%lld,\"shared_published\":%lld}",
                               g.passes, g.parallel_passes, g.rows, g.seconds, g.append_seconds, g.count_seconds, g.reserve_seconds, os, orows,
                               hts, htf, sh_n, sh_refs, sh_att, sh_pub);
    if (!js) { sqlite3_result_error_nomem(ctx); return; }
    sqlite3_result_text(ctx, js, -1, sqlite3_free);
}

#ifdef _WIN32
__declspec(dllexport)
#endif
int sqlite3_vector_init(sqlite3 *db, char **pzErrMsg, const sqlite3_api_routines *pApi) {
    /* SQLITE_EXTENSION_INIT2 is a plain store to a process global; every load_extension() of every connection repeats it with the same
     * routines table - connections opened from several threads at once would race on it with the calls that read it (TSan).  Stored once. */
    if (__atomic_load_n(&sqlite3_api, __ATOMIC_ACQUIRE) == NULL) __atomic_store_n(&sqlite3_api, pApi, __ATOMIC_RELEASE);
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
        {"vector_gpu_stats", 0, fn_gpu_stats},
        {"vector_gpu_memory", 2, fn_gpu_memory},
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
