// See tuning.h. The only getenv calls of the library live here (plus the dlopen search paths of model.cpp / the host layer).
#include "tuning.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cunvsm {

namespace {
bool env_flag(const char* name, bool dflt) {          // "0" = off, anything else = on
    const char* e = std::getenv(name);
    if (!e || !e[0]) return dflt;
    return e[0] != '0';
}
int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return (e && e[0]) ? std::atoi(e) : dflt; }
long long env_ll(const char* name, long long dflt) { const char* e = std::getenv(name); return (e && e[0]) ? std::atoll(e) : dflt; }
double env_double(const char* name, double dflt) { const char* e = std::getenv(name); return (e && e[0]) ? std::atof(e) : dflt; }
thread_local const Tuning* tl_tuning = nullptr;
}  // namespace

Tuning Tuning::from_env() {
    Tuning t;
    t.debug = env_flag("NVSM_DEBUG", t.debug);
    t.poison = env_flag("NVSM_POISON", t.poison);
    t.roctx = env_flag("NVSM_ROCTX", t.roctx);
    { const int v = env_int("NVSM_GEMM_SPLIT", t.gemm_split); t.gemm_split = (v == 6 || v == 9) ? v : 0; }
    t.gemm_rows_max = env_int("NVSM_GEMM_ROWS_MAX", t.gemm_rows_max);
    t.lazy_decay = env_flag("NVSM_LAZY_DECAY", t.lazy_decay);
    t.lazy_min_mb = env_double("NVSM_LAZY_MIN_MB", t.lazy_min_mb);
    t.dp_t_on_main = env_flag("NVSM_DP_T_ON_MAIN", t.dp_t_on_main);
    t.host_pull = env_flag("NVSM_HOST_PULL", t.host_pull);
    t.stop_events = env_flag("NVSM_STOP_EVENTS", t.stop_events);
    t.sort_layout = env_int("NVSM_SORT_LAYOUT", t.sort_layout);
    t.entry_walk_min = env_ll("NVSM_ENTRY_WALK_MIN", t.entry_walk_min);
    t.dp_fold = env_flag("NVSM_DP_FOLD", t.dp_fold);
#ifdef NVSM_EXPERIMENTS
    t.csr_grid_cap = env_int("NVSM_CSR_GRID_CAP", t.csr_grid_cap);
    { const double v = env_double("NVSM_SPLIT_RATIO", 0.0); if (v > 0.0) t.split_ratio = v; }
    t.chunk_grid_cap = env_int("NVSM_CHUNK_GRID_CAP", t.chunk_grid_cap);
    t.merged_pass = env_flag("NVSM_MERGED_PASS", t.merged_pass);
    t.chunk_blocks = env_int("NVSM_CHUNK_BLOCKS", t.chunk_blocks);
    t.row_blocks_cap = env_int("NVSM_ROW_BLOCKS_CAP", t.row_blocks_cap);
    t.entry_walk = env_flag("NVSM_ENTRY_WALK", t.entry_walk);
    t.entry_walk_min_words = env_ll("NVSM_ENTRY_WALK_MIN_WORDS", t.entry_walk_min_words);
    t.entry_walk_min_docs = env_ll("NVSM_ENTRY_WALK_MIN_DOCS", t.entry_walk_min_docs);
    t.gemm_panel = env_flag("NVSM_GEMM_PANEL", t.gemm_panel);
    { const int v = env_int("NVSM_PULL_BLOCKS", 0); if (v > 0) t.pull_blocks = v; }
    t.rows_tpw = env_int("NVSM_ROWS_TPW", t.rows_tpw);
    t.rows_dbg = env_int("NVSM_ROWS_DBG", t.rows_dbg);
    t.split_nt = env_int("NVSM_SPLIT_NT", t.split_nt);
    t.split_deal = env_int("NVSM_SPLIT_DEAL", t.split_deal);
    t.gemm_tstat = env_int("NVSM_GEMM_TSTAT", t.gemm_tstat);
    t.gemm_tstat_fwd_any = env_int("NVSM_GEMM_TSTAT_FWD_ANY", t.gemm_tstat_fwd_any);
    t.loss_epw = env_int("NVSM_LOSS_EPW", t.loss_epw);
    t.loss_pipe = env_int("NVSM_LOSS_PIPE", t.loss_pipe);
    t.csr_after = env_int("NVSM_CSR_AFTER", t.csr_after);
    t.words_csr_late = env_int("NVSM_WORDS_CSR_LATE", 0) == 1;
    t.join_e = env_int("NVSM_JOIN_E", t.join_e);
    t.split_fuse = env_flag("NVSM_SPLIT_FUSE", t.split_fuse);
    t.nt_mask = env_int("NVSM_NT", t.nt_mask);
    t.dt_on_main = env_int("NVSM_DT_ON_MAIN", t.dt_on_main);
    t.fewer_events = env_flag("NVSM_FEWER_EVENTS", t.fewer_events);
    t.docs_after_dx = env_int("NVSM_DOCS_AFTER_DX", t.docs_after_dx);
    t.docs_on_main = env_int("NVSM_DOCS_ON_MAIN", t.docs_on_main);
    t.chunk_order = env_flag("NVSM_CHUNK_ORDER", t.chunk_order);
    t.lazy_tables = env_int("NVSM_LAZY_TABLES", t.lazy_tables);
    t.aux2_prio = env_int("NVSM_AUX2_PRIO", t.aux2_prio);
    t.aux3_prio = env_int("NVSM_AUX3_PRIO", t.aux3_prio);
    t.event_fence = env_int("NVSM_EVENT_FENCE", t.event_fence);
    t.dt_slabs = env_int("NVSM_DT_SLABS", t.dt_slabs);
    t.dt_min_batch = env_int("NVSM_DT_MIN_B", t.dt_min_batch);
    t.docs_delay_us = env_int("NVSM_DOCS_DELAY_US", t.docs_delay_us);
    t.untouched_aside = env_flag("NVSM_UNTOUCHED_ASIDE", t.untouched_aside);
    t.hoist_untouched = env_int("NVSM_HOIST_UNTOUCHED", t.hoist_untouched);
    t.slab_sum_in_update = env_flag("NVSM_SLAB_SUM_IN_UPDATE", t.slab_sum_in_update);
    t.planes_in_update = env_flag("NVSM_PLANES_IN_UPDATE", t.planes_in_update);
    t.gemm_rsplit = env_flag("NVSM_GEMM_RSPLIT", t.gemm_rsplit);
    t.early_snapshot = env_flag("NVSM_EARLY_SNAPSHOT", t.early_snapshot);
    t.stamp_in_prologue = env_flag("NVSM_STAMP_IN_PROLOGUE", t.stamp_in_prologue);
    t.gather_fuse = env_int("NVSM_GATHER_FUSE", t.gather_fuse);
    t.skip_dt = env_flag("NVSM_SKIP_DT", t.skip_dt);
    t.dtw_max_batch = env_int("NVSM_DTW_MAX_B", t.dtw_max_batch);
    t.dtw_slabs = env_int("NVSM_DTW_SLABS", t.dtw_slabs);
    t.csr_fill_in_bounds = env_flag("NVSM_CSR_FILL_IN_BOUNDS", t.csr_fill_in_bounds);
#endif
    return t;
}

const char* env_raw(const char* name) { return std::getenv(name); }
bool env_bind_host() { return env_flag("NVSM_BIND_HOST", true); }

const Tuning& tuning() {
    if (tl_tuning) return *tl_tuning;
    static const Tuning process_defaults = Tuning::from_env();
    return process_defaults;
}
TuningScope::TuningScope(const Tuning* t) : prev(tl_tuning) { tl_tuning = t; }
TuningScope::~TuningScope() { tl_tuning = prev; }

const char* tuning_describe(const Tuning& t, char* buf, int n) {
    const Tuning d;
    int at = 0;
    auto add = [&](const char* name, double v, double dv) {
        if (v == dv || at >= n - 1) return;
        at += std::snprintf(buf + at, static_cast<size_t>(n - at), "%s%s=%g", at ? " " : "", name, v);
        if (at > n - 1) at = n - 1;
    };
    if (n > 0) buf[0] = 0;
    add("debug", t.debug, d.debug); add("poison", t.poison, d.poison); add("roctx", t.roctx, d.roctx);
    add("gemm_split", t.gemm_split, d.gemm_split); add("gemm_rows_max", t.gemm_rows_max, d.gemm_rows_max);
    add("lazy_decay", t.lazy_decay, d.lazy_decay); add("lazy_min_mb", t.lazy_min_mb, d.lazy_min_mb);
    add("dp_t_on_main", t.dp_t_on_main, d.dp_t_on_main); add("host_pull", t.host_pull, d.host_pull);
    add("stop_events", t.stop_events, d.stop_events); add("sort_layout", t.sort_layout, d.sort_layout);
    add("entry_walk_min", static_cast<double>(t.entry_walk_min), static_cast<double>(d.entry_walk_min));
    add("dp_fold", t.dp_fold, d.dp_fold);
#ifdef NVSM_EXPERIMENTS
    // the experiments build reads these too: every one that is off its default is named
    add("csr_grid_cap", static_cast<double>(t.csr_grid_cap), static_cast<double>(d.csr_grid_cap)); add("split_ratio", static_cast<double>(t.split_ratio), static_cast<double>(d.split_ratio)); add("chunk_grid_cap", static_cast<double>(t.chunk_grid_cap), static_cast<double>(d.chunk_grid_cap)); add("merged_pass", static_cast<double>(t.merged_pass), static_cast<double>(d.merged_pass)); add("chunk_blocks", static_cast<double>(t.chunk_blocks), static_cast<double>(d.chunk_blocks)); add("row_blocks_cap", static_cast<double>(t.row_blocks_cap), static_cast<double>(d.row_blocks_cap)); add("entry_walk", static_cast<double>(t.entry_walk), static_cast<double>(d.entry_walk)); add("entry_walk_min_words", static_cast<double>(t.entry_walk_min_words), static_cast<double>(d.entry_walk_min_words)); add("entry_walk_min_docs", static_cast<double>(t.entry_walk_min_docs), static_cast<double>(d.entry_walk_min_docs)); add("gemm_panel", static_cast<double>(t.gemm_panel), static_cast<double>(d.gemm_panel)); add("pull_blocks", static_cast<double>(t.pull_blocks), static_cast<double>(d.pull_blocks)); add("rows_tpw", static_cast<double>(t.rows_tpw), static_cast<double>(d.rows_tpw)); add("rows_dbg", static_cast<double>(t.rows_dbg), static_cast<double>(d.rows_dbg)); add("split_nt", static_cast<double>(t.split_nt), static_cast<double>(d.split_nt)); add("split_deal", static_cast<double>(t.split_deal), static_cast<double>(d.split_deal)); add("gemm_tstat", static_cast<double>(t.gemm_tstat), static_cast<double>(d.gemm_tstat)); add("gemm_tstat_fwd_any", static_cast<double>(t.gemm_tstat_fwd_any), static_cast<double>(d.gemm_tstat_fwd_any)); add("loss_epw", static_cast<double>(t.loss_epw), static_cast<double>(d.loss_epw)); add("loss_pipe", static_cast<double>(t.loss_pipe), static_cast<double>(d.loss_pipe)); add("csr_after", static_cast<double>(t.csr_after), static_cast<double>(d.csr_after)); add("words_csr_late", static_cast<double>(t.words_csr_late), static_cast<double>(d.words_csr_late)); add("join_e", static_cast<double>(t.join_e), static_cast<double>(d.join_e)); add("split_fuse", static_cast<double>(t.split_fuse), static_cast<double>(d.split_fuse)); add("nt_mask", static_cast<double>(t.nt_mask), static_cast<double>(d.nt_mask)); add("dt_on_main", static_cast<double>(t.dt_on_main), static_cast<double>(d.dt_on_main)); add("fewer_events", static_cast<double>(t.fewer_events), static_cast<double>(d.fewer_events)); add("docs_after_dx", static_cast<double>(t.docs_after_dx), static_cast<double>(d.docs_after_dx)); add("docs_on_main", static_cast<double>(t.docs_on_main), static_cast<double>(d.docs_on_main)); add("chunk_order", static_cast<double>(t.chunk_order), static_cast<double>(d.chunk_order)); add("lazy_tables", static_cast<double>(t.lazy_tables), static_cast<double>(d.lazy_tables)); add("aux2_prio", static_cast<double>(t.aux2_prio), static_cast<double>(d.aux2_prio)); add("aux3_prio", static_cast<double>(t.aux3_prio), static_cast<double>(d.aux3_prio)); add("event_fence", static_cast<double>(t.event_fence), static_cast<double>(d.event_fence)); add("dt_slabs", static_cast<double>(t.dt_slabs), static_cast<double>(d.dt_slabs)); add("docs_delay_us", static_cast<double>(t.docs_delay_us), static_cast<double>(d.docs_delay_us)); add("dt_min_batch", static_cast<double>(t.dt_min_batch), static_cast<double>(d.dt_min_batch)); add("untouched_aside", static_cast<double>(t.untouched_aside), static_cast<double>(d.untouched_aside)); add("hoist_untouched", static_cast<double>(t.hoist_untouched), static_cast<double>(d.hoist_untouched)); add("slab_sum_in_update", static_cast<double>(t.slab_sum_in_update), static_cast<double>(d.slab_sum_in_update)); add("planes_in_update", static_cast<double>(t.planes_in_update), static_cast<double>(d.planes_in_update)); add("gemm_rsplit", static_cast<double>(t.gemm_rsplit), static_cast<double>(d.gemm_rsplit)); add("early_snapshot", static_cast<double>(t.early_snapshot), static_cast<double>(d.early_snapshot)); add("stamp_in_prologue", static_cast<double>(t.stamp_in_prologue), static_cast<double>(d.stamp_in_prologue)); add("gather_fuse", static_cast<double>(t.gather_fuse), static_cast<double>(d.gather_fuse)); add("skip_dt", static_cast<double>(t.skip_dt), static_cast<double>(d.skip_dt)); add("dtw_max_batch", static_cast<double>(t.dtw_max_batch), static_cast<double>(d.dtw_max_batch)); add("dtw_slabs", static_cast<double>(t.dtw_slabs), static_cast<double>(d.dtw_slabs)); add("csr_fill_in_bounds", static_cast<double>(t.csr_fill_in_bounds), static_cast<double>(d.csr_fill_in_bounds));
    at += std::snprintf(buf + at, static_cast<size_t>(n - at), "%s(experiments build)", at ? " " : "");
#endif
    return buf;
}

}  // namespace cunvsm
