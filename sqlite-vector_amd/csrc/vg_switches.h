// vg_switches.h - every environment switch of the engine, read in ONE place (vg_switches_read, vg_corpus.hip) into one process-wide table:
// when a corpus / a shard set is created and when a test asks (vg_reload_switches, vectorgpu_diag.h) - never on a query's path (round 4
// read ~60 getenv sites, some of them per enqueue).  Switches marked LAB exist in -DVG_LAB builds only (tools/build_*_variants.sh): they
// are A/B switches whose losing side docs/EXPERIMENTS.md records; a product build compiles them to "unset".
#pragma once
#include <atomic>
#include <climits>

enum VgSwitch {
    SW_VECTORGPU_SHARD_GATHER,
    SW_VECTORGPU_SHARD_THREADS,
    SW_VG_BATCH_BPC,                       // LAB
    SW_VG_BATCH_H_SPLIT,
    SW_VG_BATCH_H_WAVES,
    SW_VG_BATCH_LONG,
    SW_VG_BATCH_LONG_BPC,                       // LAB
    SW_VG_BATCH_MFMA,
    SW_VG_BATCH_MIN_QUERIES,                       // LAB
    SW_VG_BATCH_PREPASS,                       // LAB
    SW_VG_BATCH_Q8,
    SW_VG_BATCH_SLICE,
    SW_VG_BATCH_STAGES,
    SW_VG_BATCH_TILE_MAJOR,
    SW_VG_BLOCKS_PER_CU,                       // LAB
    SW_VG_F32_FILTER,
    SW_VG_FILTER_LPR_LOG2,                       // LAB
    SW_VG_FILTER_U,                       // LAB
    SW_VG_FORCE_LONG,
    SW_VG_HALF_COSN,                       // LAB
    SW_VG_HOST_DIRECT,                       // LAB
    SW_VG_KEYS_DIRECT,                       // LAB
    SW_VG_LPR_LOG2,                       // LAB
    SW_VG_MULTI_SCAN,
    SW_VG_NT,                       // LAB
    SW_VG_Q8_TWO_READS,                       // LAB
    SW_VG_RADIX_SELECT,
    SW_VG_REF_ALWAYS_EMIT,                       // LAB
    SW_VG_REF_STORE_MODE,
    SW_VG_SCAN_FILTER,
    SW_VG_SCAN_FILTER_MIN_MB,
    SW_VG_SCAN_FILTER_MIRROR_COPY,                       // LAB
    SW_VG_SCAN_FILTER_N4,
    SW_VG_SCAN_FILTER_NO_GUARD,
    SW_VG_SCAN_FILTER_PREMERGE,                       // LAB
    SW_VG_SCAN_FILTER_PREPASS,                       // LAB
    SW_VG_SCAN_FILTER_PREPASS_DIV,                       // LAB
    SW_VG_SCAN_FILTER_SHADOW,
    SW_VG_SCAN_ORDER,                       // LAB
    SW_VG_SHAPE_BF16_L2_U3,                       // LAB
    SW_VG_SHAPE_F16_ROUND3,                       // LAB
    SW_VG_SHAPE_INT_SHORT_ROUND3,                       // LAB
    SW_VG_SHAPE_PREF_ROUND1,                       // LAB
    SW_VG_U,                       // LAB
    VGSW_COUNT
};
#define VGSW_UNSET INT_MIN
// (atomics, relaxed: a corpus created on one thread re-reads the environment while scans of other connections read the table - the same
//  values nearly always, but a plain int array made that a data race)
extern std::atomic<int> vg_switch_values[VGSW_COUNT];          // VGSW_UNSET: not in the environment (vg_corpus.hip)
void vg_switches_read(void);                      // (re)reads the environment
// the switch's integer value, dflt when it is unset (VG_SCAN_FILTER_SHADOW: its first letter; VECTORGPU_SHARD_GATHER: 1 = "rccl")
static inline int vg_sw(VgSwitch id, int dflt) { const int v = vg_switch_values[id].load(std::memory_order_relaxed); return v == VGSW_UNSET ? dflt : v; }
static inline bool vg_sw_set(VgSwitch id) { return vg_switch_values[id].load(std::memory_order_relaxed) != VGSW_UNSET; }
