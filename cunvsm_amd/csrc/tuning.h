// Every switch of the engine in ONE place, read from the environment ONCE per handle (nvsm_create) — never on a launch path.
//   * the documented switches (each one exercised by a test) are read by every build;
//   * the experiment switches — launch shapes, stream placement, kernel choice: what A/B runs turn — are read only by builds
//     with -DNVSM_EXPERIMENTS (`make dbg` → libcunvsm_amd_dbg.so); the shipped library runs their defaults, so a stray
//     variable in a user's environment cannot change what it does.
#pragma once

namespace cunvsm {

struct Tuning {
    // ---- documented --------------------------------------------------------------------------------------------------------
    bool debug = false;                 // NVSM_DEBUG=1            finite checks of every intermediate + a sync per call
    bool poison = false;                // NVSM_POISON=1           uncleared device buffers start as 0xFF bytes
    bool roctx = true;                  // NVSM_ROCTX=0            no roctx ranges (process-wide: read at the first use)
    int gemm_split = 6;                 // NVSM_GEMM_SPLIT=0|6|9   partial products of the split-bf16 projection kernels (0: exact-fp32 MFMA kernels)
    int gemm_rows_max = 8192;           // NVSM_GEMM_ROWS_MAX=n    largest batch of the row-panel projection kernel
    bool lazy_decay = true;             // NVSM_LAZY_DECAY=0       eager dense decay for every table
    double lazy_min_mb = -1.0;          // NVSM_LAZY_MIN_MB=x      table state from which decay is lazy (< 0: 96 MB sparse Adam, 384 MB otherwise)
    bool dp_t_on_main = false;          // NVSM_DP_T_ON_MAIN=1     data parallel: all three collectives of a step on the main stream
    bool host_pull = true;              // NVSM_HOST_PULL=0        page-locked batches through hipMemcpyAsync instead of the pull kernel
    bool stop_events = true;            // NVSM_STOP_EVENTS=0      plain event records instead of events riding on kernel launches
    int sort_layout = -1;               // NVSM_SORT_LAYOUT=0..4   which side streams build the two CSRs (< 0: by batch size)
    long long entry_walk_min = 64ll * 4096;      // NVSM_ENTRY_WALK_MIN=n   entries from which a split table pass walks the sorted entries
    bool dp_fold = true;                // NVSM_DP_FOLD=0          data parallel without synchronised batch-norm: [db | loss] in an f64 all-reduce of their own instead of behind dT
    // ---- experiments (-DNVSM_EXPERIMENTS) ------------------------------------------------------------------------------------
    int csr_grid_cap = -1;              // NVSM_CSR_GRID_CAP
    double split_ratio = 2.0;           // NVSM_SPLIT_RATIO
    int chunk_grid_cap = 0;             // NVSM_CHUNK_GRID_CAP
    bool merged_pass = true;            // NVSM_MERGED_PASS
    int chunk_blocks = 2048;            // NVSM_CHUNK_BLOCKS
    int row_blocks_cap = 256 * 32;      // NVSM_ROW_BLOCKS_CAP
    bool entry_walk = true;             // NVSM_ENTRY_WALK
    long long entry_walk_min_words = -1, entry_walk_min_docs = -1;      // NVSM_ENTRY_WALK_MIN_WORDS / _DOCS
    bool gemm_panel = true;             // NVSM_GEMM_PANEL
    int pull_blocks = 8;                // NVSM_PULL_BLOCKS
    int rows_tpw = 0;                   // NVSM_ROWS_TPW
    int rows_dbg = 0;                   // NVSM_ROWS_DBG
    int split_nt = 0;                   // NVSM_SPLIT_NT
    int split_deal = 0;                 // NVSM_SPLIT_DEAL
    int gemm_tstat = 3;                 // NVSM_GEMM_TSTAT
    int gemm_tstat_fwd_any = -1;        // NVSM_GEMM_TSTAT_FWD_ANY
    int loss_epw = 0;                   // NVSM_LOSS_EPW
    int loss_pipe = -1;                 // NVSM_LOSS_PIPE (two row sets per wave in the loss kernel: -1 by rule, 0 off, 1 on)
    int csr_after = 0;                  // NVSM_CSR_AFTER
    bool words_csr_late = false;        // NVSM_WORDS_CSR_LATE
    int join_e = 0;                     // NVSM_JOIN_E
    bool split_fuse = true;             // NVSM_SPLIT_FUSE
    int nt_mask = 3;                    // NVSM_NT
    int dt_on_main = -1;                // NVSM_DT_ON_MAIN
    bool fewer_events = true;           // NVSM_FEWER_EVENTS
    int docs_after_dx = -1;             // NVSM_DOCS_AFTER_DX
    int docs_on_main = 0;               // NVSM_DOCS_ON_MAIN
    bool chunk_order = true;            // NVSM_CHUNK_ORDER
    int lazy_tables = 3;                // NVSM_LAZY_TABLES
    int aux2_prio = 0, aux3_prio = 1;   // NVSM_AUX2_PRIO / NVSM_AUX3_PRIO
    int event_fence = -1;               // NVSM_EVENT_FENCE
    int dt_slabs = 0;                   // NVSM_DT_SLABS (0: by batch size)
    int docs_delay_us = 0;              // NVSM_DOCS_DELAY_US (a spin kernel in front of the documents update on its side stream)
    int dt_min_batch = 40960;           // NVSM_DT_MIN_B (the split-bf16 dT kernel from this batch size up)
    bool untouched_aside = true;        // NVSM_UNTOUCHED_ASIDE
    int hoist_untouched = 2;           // NVSM_HOIST_UNTOUCHED (fused step: the decay of the words rows without entries behind the CSR build: 0 off, 1 in front of the build's event, 2 behind it)
    bool slab_sum_in_update = true;     // NVSM_SLAB_SUM_IN_UPDATE (fused step: the projection update adds up the dT product's slabs; 0: launch_splitk_reduce)
    bool planes_in_update = true;       // NVSM_PLANES_IN_UPDATE (the projection update writes T's bf16 planes itself; 0: two launches behind it)
    bool gemm_rsplit = true;            // NVSM_GEMM_RSPLIT (the split-bf16 row-panel kernel at per-rank batch sizes; 0: gemm_rows)
    bool early_snapshot = true;         // NVSM_EARLY_SNAPSHOT (words scalar snapshot behind the CSR build)
    bool stamp_in_prologue = true;      // NVSM_STAMP_IN_PROLOGUE (words stamps set by the next step's prologue)
    int dtw_max_batch = 16383;          // NVSM_DTW_MAX_B (the wave-sized dT kernel, gemm_dtw.hip, up to this batch where gemm_dt does not run; 0: never — the tiled exact-fp32 kernel)
    int dtw_slabs = 0;                  // NVSM_DTW_SLABS (0: the slabs of the tiled kernel, ~400 rows each)
    bool csr_fill_in_bounds = false;    // NVSM_CSR_FILL_IN_BOUNDS=1 (small batches: the bounds kernel also writes the long rows' chunk descriptors — one launch less, bit-identical; measured LSE +3 %, batch 6 400 / 3 200 ±0: NOTES_r06 §3 — hence off)
    bool skip_dt = false;               // NVSM_SKIP_DT (TIMING ONLY, wrong results: the dT product is not launched — is it on the step's critical path?)
    int gather_fuse = 0;                // NVSM_GATHER_FUSE=1 (the word gather-mean inside the per-rank forward product's staging, gemm_rsplit.hip GATH: bit-identical, measured 2-3.5 % SLOWER per step — profiles/NOTES_r06.md §1 — hence off)

    static Tuning from_env();
};

// The switches in force for the calling thread: the handle's inside a call on a handle (Model's entry points install them),
// the values of a debug hook's own from_env() inside a debug hook, otherwise the process defaults (read once).
const Tuning& tuning();
struct TuningScope {
    const Tuning* prev;
    explicit TuningScope(const Tuning* t);
    ~TuningScope();
    TuningScope(const TuningScope&) = delete;
    TuningScope& operator=(const TuningScope&) = delete;
};
const char* env_raw(const char* name);      // the value of a variable that is not ours (HIP_VISIBLE_DEVICES, ROCR_VISIBLE_DEVICES: nvsm_bind_host_thread), or null
bool env_bind_host();      // NVSM_BIND_HOST=0: nvsm_bind_host_thread leaves the calling thread's affinity alone (read per call)
const char* tuning_describe(const Tuning& t, char* buf, int n);      // "name=value ..." of every switch that differs from its default

}  // namespace cunvsm
