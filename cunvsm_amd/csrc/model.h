// Host side of the MI355X NVSM / LSE engine: owns the parameters, optimiser state and per-step
// workspaces in HBM and sequences the gfx950 kernels of kernels.h on one HIP stream.
// Mirrors Model<TextEntity::Objective> (include/cuNVSM/model.h:75-131): initialize / compute_cost /
// compute_gradients / update / get_cost, same argument meaning.
#pragma once

#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/cunvsm_amd.h"
#include "kernels.h"

namespace cunvsm {

#define NVSM_HIP_CHECK(expr)                                                                         \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            throw ::cunvsm::Error(NVSM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline bool devbuf_poison() { return tuning().poison; }

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void alloc(size_t count, bool zero = false) {
        release();
        n = count;
        if (count == 0) return;
        NVSM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        // NVSM_POISON=1 (tests): buffers that are not cleared start as 0xFF bytes (NaN floats, -1 ints) instead of whatever
        // the allocator hands back — a kernel that reads one before it is written shows up at once
        if (!zero && !devbuf_poison()) return;
        NVSM_HIP_CHECK(hipMemset(p, zero ? 0 : 0xFF, count * sizeof(T)));
        // hipMemset of device memory returns once the fill is queued on the null stream, and the handle's streams are
        // non-blocking (no implicit order with the null stream): without this wait a kernel of the handle could read the
        // buffer before it was cleared (seen as a memory fault of lazy_refresh_kernel on a recycled, not yet zeroed stamp array)
        NVSM_HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// RCCL entry points resolved at run time (dlopen) so that single-GPU use never loads librccl and a
// process that already holds PyTorch's RCCL shares it.
struct RcclApi;

class Profiler {
 public:
    bool enabled = false;
    std::string only;          // when not empty, only these kernel groups (comma-separated) are timed (a few events per step instead of ~50)
    void begin(const char* name, hipStream_t s);
    void end(hipStream_t s);
    // the next event pair of group `name`, NOT recorded: the caller hands it to a single launch as that kernel's start / stop
    // events (kernels.h set_launch_events) — the group's time is then the kernel's own execution time, and nothing is queued
    // in front of or behind the kernel. false: the group is not being timed.
    bool bind(const char* name, hipEvent_t* start, hipEvent_t* stop);
    bool selected(const char* name) const;
    void note(const char* name) { if (enabled) ++notes_[name]; }      // a counter without events: which code path a launch took
    void reset();
    std::vector<std::string> names() const;
    bool get(const std::string& name, double* ms, int64_t* launches);
    ~Profiler();
 private:
    struct Slot { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; };
    std::map<std::string, Slot> slots_;
    std::map<std::string, int64_t> notes_;
    Slot* cur_ = nullptr;
};

struct TableState {           // one embedding table + its optimiser state + its per-step CSR workspace
    int64_t rows = 0;
    int dim = 0;
    DevBuf<float> P, m, vfull, sc[2];
    int sc_cur = 0;
    uint64_t t = 1;           // Adam step counter (cpp/updates_adam.cu:130)
    // CSR workspace. What the build writes and the update reads comes in two sets for the documents table: the build
    // of step k+1 (side stream, at the start of the step) then does not have to wait for the documents update of step k,
    // which trails into step k+1 and still reads the arrays of step k.
    struct CsrIndex {
        DevBuf<int> sorted_key, sorted_entry, chunk_base, chunk_desc, chunk2_base, chunk2_desc;
        DevBuf<int> touched;      // rows with entries (Csr::touched)
        DevBuf<int> chunk_order;  // Csr::chunk_order (large batches only)
        DevBuf<int> csr_zeroed;   // [row_begin (rows) | row_end (rows) | num_chunks (2) | num_touched | pad]: cleared by the sort's first launch
    };
    CsrIndex idx[2];
    int idx_sets = 1, idx_cur = 0;
    DevBuf<int> arrive_row, arrive2;      // arrival counters of the one-launch table pass (Csr)
    DevBuf<float> partial, partial_q, partial2, partial2_q;
    DevBuf<int> chunk_key, chunk_key_sorted;      // scratch of launch_chunk_order
    DevBuf<char> sort_temp;
    size_t sort_temp_bytes = 0;
    uint64_t sort_epoch = 0;      // (unused by the two-launch sort)
    int sort_bits = 1;
    int max_chunks = 0, max_chunks2 = 0;
    int64_t max_entries = 0;
    // lazy dense decay (kernels.h): for tables with at least as many rows as a batch has entries
    bool lazy = false;
    bool lazy_scalar = false;     // the per-row scalar state takes part (Adam v, document Adagrad accumulator)
    DevBuf<int> stamp;            // [rows] updates applied to the row
    int updates_done = 0;
    float decay_hist[kLazyHistory];
};

class Model {
 public:
    explicit Model(const nvsm_config& cfg);
    ~Model();

    void initialize(uint64_t seed);
    void initialize_from_rng_state();      // Glorot draws continue from the generator's current state
    uint64_t rng_get_state();
    void rng_set_state(uint64_t s);

    void compute_cost(const nvsm_batch& batch, const int64_t* entity_ids);
    void compute_gradients();
    void update(float lr, float scaled_lambda);
    float get_cost();
    double cost_f64() const { return cost_; }      // valid after get_cost()
    float scaled_regularization_lambda() const;
    void step(const nvsm_batch& batch, const int64_t* entity_ids, float lr, float* cost);

    int64_t param_size(const std::string& name);
    void get_param(const std::string& name, float* dst, int64_t count);
    void set_param(const std::string& name, const float* src, int64_t count);
    void increment_param(const std::string& name, int64_t index, float delta);
    int64_t tensor_size(const std::string& name);
    void get_tensor(const std::string& name, float* dst, int64_t count);

    void set_stream(hipStream_t s);
    int64_t step_deferred(const nvsm_batch& batch, const int64_t* entity_ids, float lr);
    float deferred_cost(int64_t ticket);
    void wait_inputs();
    void synchronize();
    void debug_delay(int microseconds);
    void draw_reference_negatives(const int64_t* labels, int64_t B, int64_t* ids);
    void join_T();
    void join_E();
    void join_aux() { join_T(); join_E(); }
    void comm_init(const char id[128]);
    void average_tables();                 // data parallel: mean of the replicas' embedding tables (nvsm_dp_average_tables)
    int comm_ranks() const { return comm_ranks_; }
    bool dp_single_stream() const { return dp_single_stream_; }
    void set_allreduce_callback(nvsm_allreduce_fn fn, void* user) { ar_fn_ = fn; ar_user_ = user; }

    Profiler prof;
    const nvsm_config& config() const { return cfg_; }
    const Tuning& tune() const { return tune_; }
    std::string describe(int64_t batch) const;      // which kernel each product of a step takes at `batch` windows, table modes, switches off their defaults

 private:
    struct ParamRef { float* p; int64_t n; };
    ParamRef find_param(const std::string& name);
    void build_csr(TableState& t, const int* keys, int64_t n, hipStream_t s);
    Csr csr_of(TableState& t, int64_t n);
    void backward_dx();                                  // B5, B7, B9 on the main stream
    void backward_T(hipStream_t s);                      // B6 (+ its all-reduce)
    void update_entities(float lr, float sl, hipStream_t s, hipEvent_t row_pass_after = nullptr);
    void update_words(float lr, float sl);
    void update_transform(float lr, float sl, hipStream_t s);
    void allreduce_f64(double* dev, int64_t n);
    void allreduce_f32(float* dev, int64_t n, hipStream_t s);
    void lazy_refresh(TableState& t, const Csr* touched, hipStream_t s);     // null = every row of the table
    void lazy_flush_all();                 // before anything reads or writes whole tables (get / set_param, averaging)
    LazyView lazy_view(const TableState& t) const;      // what a reader needs to bring the rows it gathers up to date on the fly
    void lazy_scalar_snapshot(TableState& t, const Csr& c, hipStream_t s);   // the touched rows' scalar state, before an update's passes
    void lazy_begin_update(TableState& t, RowPassArgs& a, bool scalar_pingpong);
    void lazy_end_update(TableState& t, const Csr& c, hipStream_t s);
    RowPassArgs final_words_pass_args(float lr, float sl);
    bool hoist_untouched_ = false;        // step() -> compute_cost: queue the words rows-without-entries decay behind the CSR build
    float hoist_lr_ = 0.f, hoist_sl_ = 0.f;
    bool words_untouched_hoisted_ = false;      // ... done for this step's update
    bool words_untouched_pending_ = false;      // ... and not yet followed by the main stream (the next word gather does)
    hipEvent_t ev_untouched_ = nullptr;
    void settle_words_stamp();            // the words table's pending stamps, now (see lazy_end_update)
    bool words_stamp_pending_ = false;    // the last words update's stamps have not been set yet
    int64_t words_stamp_n_ = 0;           //   ... entries of that update's CSR
    bool words_snapshot_early_ = false;   // this step's words scalar snapshot was taken behind the CSR build
    void raise_device_error();             // throws when a kernel has flagged bad ids / non-finite values since the last check
    void debug_check(const float* x, int64_t n, int which);
    void alloc_table(TableState& t, int64_t rows, int dim, int64_t max_entries);
    float adam_bc(uint64_t t) const;

    Tuning tune_;                     // the switches of this handle: read from the environment once, by the constructor (tuning.h)
    nvsm_config cfg_;
    int R_;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = false;
    // The batch → CSR builds depend only on the indices, so they run on a side stream concurrently with the
    // forward / backward kernels and are joined right before the row passes.
    hipStream_t aux_stream_ = nullptr;
    hipStream_t aux2_stream_ = nullptr;   // the words CSR build: next to the documents CSR build instead of behind it
    hipStream_t aux3_stream_ = nullptr;   // host-batch copies, then the documents CSR build (NVSM_SORT_LAYOUT 4)
    hipEvent_t ev_csr_ents_ = nullptr;
    hipEvent_t ev_inputs_ = nullptr, ev_csr_ = nullptr;
    // fused step(): the documents update (HBM bound) and the dT GEMM (MFMA bound) run on the side stream next to the
    // dx GEMM and the words update on the main stream
    hipEvent_t ev_gathered_ = nullptr;
    hipEvent_t ev_loss_ = nullptr, ev_dx_ = nullptr, ev_bwdx_ = nullptr, ev_E_done_ = nullptr, ev_T_done_ = nullptr, ev_words_late_ = nullptr;
    hipEvent_t ev_cost_ready_ = nullptr, ev_cost_copied_ = nullptr;      // step(cost): the loss word copied out behind the loss kernel
    double* cost_host_ = nullptr;                                         // ... into this page-locked word
    hipStream_t words_csr_stream_ = nullptr;            // the side stream that built this step's words CSR (NVSM_SORT_LAYOUT)
    hipStream_t words_untouched_stream_ = nullptr;      // set by step() around update_words (kernels.h launch_table_pass untouched_s)
    bool words_tail_pending_ = false;                   // side stream 2 still decays words rows: the next word gather joins it
    hipStream_t dx_follower_ = nullptr;                 // set by step(): the stream that waits for ev_dx_ (issued inside backward_dx)
    hipEvent_t loss_stop_event_ = nullptr;      // set by step(): the loss kernel of this compute_cost carries it as its completion event
    std::minstd_rand0 rng_;           // include/cuNVSM/base.h:36
    uint64_t device_seed_ = 1, step_count_ = 0;

    TableState words_, ents_;
    DevBuf<float> T_, b_, s0T_, s0b_, s1T_, s1b_;
    uint64_t t_transform_ = 1;

    // per-step inputs
    DevBuf<int64_t> in_words_[2], in_labels_[2], in_ids64_;      // host batches: two staging sets (see compute_cost)
    DevBuf<float> in_wwts_[2], in_instw_[2];
    int in_parity_ = 0;
    hipStream_t copy_stream_ = nullptr;   // alias of aux3_stream_
    hipEvent_t ev_copied_ = nullptr, ev_step_begin_[2] = {nullptr, nullptr};
    bool copied_recorded_ = false, last_batch_on_host_ = false;
    DevBuf<int> widx_, ids_buf_[2];
    int* ids_p_ = nullptr;            // this step's document ids: the two buffers alternate, so that a step's prologue never
                                      //   rewrites the ids the previous step's documents CSR build may still be reading
    const float* wwts_ = nullptr;     // device pointer or null
    const float* instw_ = nullptr;
    const int64_t* labels_dev_ = nullptr;
    std::vector<int64_t> host_labels_;
    // host sampler: two page-locked id buffers, each guarded by the event of the copy that last read it, so that
    // compute_cost never waits for the stream (the step before last has long released the buffer it reuses)
    int64_t* host_ids_pin_[2] = {nullptr, nullptr};
    hipEvent_t ev_host_ids_[2] = {nullptr, nullptr};
    bool host_ids_used_[2] = {false, false};
    int host_ids_parity_ = 0;
    int* err_host_ = nullptr;         // page-locked error word written by kernels (kernels.h: NVSM_BAD_*, NVSM_NONFINITE_BASE)
    bool debug_ = false;              // NVSM_DEBUG=1: finite checks of every intermediate + a sync after each call
    int64_t B_ = 0;                   // instances of the current batch (this rank)

    // intermediates
    DevBuf<float> phrase_raw_, phrase_norms_, ge_msq_;      // optional L2 normalisers: cached raw phrase means + norms; per-entry mean of squares
    DevBuf<float> phrase_alt_;            // second phrase matrix (see compute_cost)
    float* phrase_p_ = nullptr;           // the one the current forward result lives in
    bool E_pending_ = false, T_pending_ = false;      // side-stream tails of the last nvsm_step not yet joined
    DevBuf<float> phrase_, pre_, proj_, dy_, gphrase_, coef_, probs_, pp_, msq_w_, msq_parts_, U_, scale_w_, grad_entity_;
    DevBuf<double> stats_;                   // [2 de | 1 + 2 de] = Σx Σx² | loss Σdy Σdy·x̂ — written by the ordered grid sums
    // workspaces of those sums (kernels.h GridSumWs): projection GEMM epilogue / loss kernel
    struct SumsBufs { DevBuf<float> part; DevBuf<double> part2; DevBuf<int> arrive; GridSumWs ws{}; };
    SumsBufs sums_fwd_, sums_bwd_;
    // the projection matrix cut into bf16 planes for the split-bf16 GEMM (gemm_split.hip), in the forward and the backward
    // product's layout; `ready` is cleared by everything that writes T
    DevBuf<char> planes_fwd_, planes_bwd_, rplanes_fwd_, rplanes_bwd_;
    bool fuse_slab_sum_ = false;         // step(): backward_T leaves the slab sum to the projection update behind it
    int pending_slabs_ = 0;              //   ... that many slabs in gT_partial_
    void planes_stale();                 // T changed: every set of planes is out of date
    GemmSplitWs split_fwd_{}, split_bwd_{};
    void cut_transform_planes(hipStream_t strm);
    bool dt_on_main() const;
    bool dt_on_main_at(int64_t B) const;
    bool use_dt_at(int64_t B) const;
    bool use_dtw_at(int64_t B) const;    // ... on the wave-sized kernel (gemm_dtw.hip: per-rank batches)
    int csr_stream_layout() const;
    bool slab_sum_fusable() const;     // the projection update's own slab sum is launch_splitk_reduce's (vector order, alignment)
    hipStream_t words_untouched_stream_prev_ = nullptr;      // the stream the hoisted words decay of the last step ran on (hoist_untouched == 2)
    bool dp_fold() const;              // data parallel: [db | loss] ride on the dT all-reduce (one collective per step)
    bool gather_fused_at(int64_t B) const;      // the forward product at this batch size forms the phrase rows itself
    int last_csr_layout_ = -1;          // the layout of the previous step's builds (host-batch copies lean on it)
    void alloc_sums(SumsBufs& b, int colgroups, int contrib_cap, int width_cap);
    bool csr_joined_words_ = true, csr_joined_ents_ = true;      // the main stream is behind the current CSR builds
    double* stats_fwd_ = nullptr;
    double* stats_bwd_ = nullptr;
    DevBuf<float> bn_mean_, bn_inv_std_, dbeta_, dgamma_;
    DevBuf<float> gT_, gb_, gT_partial_;
    int gemm_slabs_want_ = 128;        // split-K slabs of the exact-fp32 dT product (shapes / settings gemm_dt.hip does not cover)
    bool dt_ok_ = false;               // the split-K dT kernel (gemm_dt.hip) covers this model's shapes
    int num_cus_ = 256;
    bool table_decays_lazily(bool documents, int64_t rows, int dim, int64_t max_entries) const;
    static constexpr int kDtwSlabs = 12;                    // split-K slabs of the wave-sized dT kernel (gemm_dtw.hip; backward_T)
    static constexpr int64_t kDtMainMinBatch = 16384;      // eager tables, one rank: the dT product on the split-bf16 kernel, on the main stream, from here
    int chunk_entries(const TableState& t, int64_t n) const;      // entries per level-1 chunk of a long row for a batch of n entries of table t
    bool use_dt() const;               // this step's dT product runs on it (else: the exact-fp32 tiled / panel kernels)

    bool have_forward_ = false, have_grads_ = false;
    struct DeferredCost { double* host = nullptr; hipEvent_t ev = nullptr; double batch = 1.0; int64_t ticket = -1; };
    DeferredCost deferred_[NVSM_MAX_DEFERRED];
    int64_t next_ticket_ = 0;
    bool inputs_recorded_ = false;
    double cost_ = 0.0;
    bool cost_valid_ = false;
    bool loss_reduced_ = false;       // data parallel: the loss word has been all-reduced (by the backward pass)

    // data parallel
    RcclApi* rccl_ = nullptr;
    void* comm_ = nullptr;
    int comm_ranks_ = 0;              // ncclCommCount of the communicator (0 = none)
    // the step's three collectives: on two event-ordered streams (dT all-reduce + projection update on side stream 2, next to
    // the words update) or all on the main stream — NVSM_DP_T_ON_MAIN=1, or chosen by comm_init when the communicator does not
    // return the right sums with the two-stream order
    bool dp_single_stream_ = false;
    bool comm_order_check(bool two_streams);      // all-reduces of known values in the step's order; every rank gets the same verdict
    DevBuf<double> loss_tmp_;         // data parallel get_cost before compute_gradients: all-reduced copy of the loss word
    DevBuf<double> loss_red_;         // ... with the folded collective (dp_fold): the summed loss word, written behind the dT all-reduce
    bool loss_folded_ = false;        //     this step's summed loss is in loss_red_, final on loss_stream_
    hipStream_t loss_stream_ = nullptr;
    nvsm_allreduce_fn ar_fn_ = nullptr;
    void* ar_user_ = nullptr;
    std::vector<double> ar_host_;

    // Data parallel with exact tables (nvsm_config.dp_exact_tables): what the table updates read — ids, projected phrases,
    // multipliers, phrase gradients — of ALL ranks, rank-major, all-gathered on the main stream; the CSR builds and the table
    // passes then run on world_size x B windows, i.e. every rank performs the single-GPU update of the global batch.
    bool exact_ = false;
    struct UpdateInputs {
        const float* proj; const float* coef; const float* pp; const int* ids; const float* gphrase; const float* wwts;
        const float* msq_w; const int* widx; int64_t B;
    };
    UpdateInputs update_inputs() const;      // the rank's own buffers, or the gathered ones
    DevBuf<int> xg_ids_[2], xg_widx_;
    int* xg_ids_p_ = nullptr;                // alternates like ids_p_
    DevBuf<float> xg_proj_, xg_coef_, xg_pp_, xg_gphrase_, xg_msq_w_, xg_wwts_;
    void allgather(const void* send, void* recv, size_t bytes, hipStream_t s);      // recv: world_size x bytes, rank-major
    void gather_update_inputs();
};

}  // namespace cunvsm
