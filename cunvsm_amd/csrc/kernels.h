// Launchers of the hand-written gfx950 kernels of the NVSM / LSE hot path. Host-callable; every
// launcher enqueues on the given stream and returns immediately. See DESIGN.md for the data layout
// and the per-kernel roofline; each kernel cites the reference code it replaces.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

#include "tuning.h"

namespace cunvsm {

// what every layer below the C ABI throws; c_api.cpp turns it into an nvsm_status + nvsm_last_error() (status: NVSM_ERR_*)
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// An event recorded behind a kernel is a packet of its own which the stream's next kernel waits for: a bubble of 6-10 us
// per record on the step's critical stream. A kernel launched through NVSM_LAUNCH right after set_launch_events(start, stop)
// carries them as its own start / completion events instead (hipExtLaunchKernelGGL): `stop` means the same for whoever waits
// on it as a record behind the kernel, and a (start, stop) pair made with timing brackets exactly the kernel's execution —
// no packet between the kernel and its neighbours either way. take_launch_events() hands the pending events to the launch
// (or back to the caller, who records them the plain way when the launcher launched nothing). Per host thread.
void set_launch_events(hipEvent_t start, hipEvent_t stop);
hipEvent_t take_launch_events(hipEvent_t* start);      // returns the stop event
#define NVSM_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                       \
    do {                                                                                                          \
        hipEvent_t _start = nullptr;                                                                              \
        hipEvent_t _stop = ::cunvsm::take_launch_events(&_start);                                                 \
        if (_stop || _start)                                                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, _start, _stop, 0, __VA_ARGS__);             \
        else                                                                                                      \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                  \
    } while (0)

// How a reader brings the rows of a lazily decayed table up to date while it gathers them (see "lazy dense decay" below):
// row r has had `stamp[r]` updates applied, the table `now`; the factors of the updates in between are applied to the
// loaded values one by one, in update order — the roundings of the dense passes — without writing anything back.
constexpr int kLazyHistory = 128;
struct LazyView {
    const int* stamp;               // null: the table is current (no lazy decay)
    int now;
    float decay[kLazyHistory];      // factor of update u + 1 on the table rows at [u % kLazyHistory]
};

// Workspace of the ordered grid-wide sums (device_utils.h grid_sum_ordered): the batch-norm column statistics out of the
// projection GEMM's epilogue and [loss | Σdy | Σdy·x̂] out of the loss kernel are added up in a fixed order — same bits
// every run, no atomics on data. `colgroups` independent sums (column parts / tiles of a GEMM; 1 for the loss kernel), each
// of up to `contrib_cap` contributions of up to `width_cap` floats.
struct GridSumWs {
    float* part;        // [colgroups][contrib_cap][width_cap]
    double* part2;      // [colgroups][groups_cap][width_cap]
    int* arrive;        // [colgroups][groups_cap + 1], zero between launches
    int colgroups, contrib_cap, groups_cap, width_cap, fan;
};
// The bf16 planes of the new T for the next step's projection products, written by the update itself (a thread holds the new
// element anyway: two launches of 5 us less behind the update, on the stream the next forward product waits for).
// kind 0: none; 1: gemm_split.hip's layout (dim = padded columns np); 2: gemm_rsplit.hip's (dim = 32-column tiles).
// transposed 0: the forward product's B (k = row of the stored T = word dimension, n = entity dimension); 1: the backward one's.
struct PlaneTarget { unsigned char* planes; size_t plane_stride; int kind; int dim; int transposed; };
// batch-norm backward (or, pre == null, the bias gradient alone) riding on the backward projection product: see launch_gemm_rows
struct BnDxFused { float* dy; const float* pre; const float* mean; const float* inv_std; const double* sums;
                   float* dbeta; float* dgamma; float* grad_bias; double n_global; };
// the word gather-mean (launch_gather_mean's arguments) riding on the forward projection product's staging of A (gemm_rsplit.hip,
// gemm_split.hip): phrase[b] = (Σ_j wts[b, j] · table[idx[b, j]]) / window is formed as the product stages its rows and written
// out on the way, bit for bit what the gather kernel writes
struct GatherFused { const float* table; const int* idx; const float* wts; int window; const LazyView* lazy; };
// planes of the small operand for the split-bf16 GEMM (gemm_split.hip; see launch_gemm_split)
// ... and for the split-bf16 row-panel kernel of the per-rank batch sizes (gemm_rsplit.hip), which wants them in another order
struct GemmSplitWs { void* planes; size_t bytes; bool ready; void* rplanes; size_t rbytes; bool rready; };
inline int grid_sum_fan(int contributions) { return contributions > 256 ? 32 : 16; }

// ---- gather-mean (F3/F9; replaces average_repr_kernel, cpp/params.cu:75-95) ------------------
void launch_gather_mean(const float* table, int dim, const int* idx, const float* wts, int window,
                        int64_t num_out, float* out, hipStream_t s, const LazyView* lazy = nullptr);

int window_unroll(int window);      // rows of a window in flight per lane in the window-major gathers (gather-mean, adam_u)

// ---- fp32 MFMA GEMM (F5, B6, B7; replaces the cuBLAS calls at cpp/params.cu:417,528, objective.cu:453)
// C[M][N] = alpha * A·B (+ bias[n]).  a_layout 0: A is [M][K] (lda); 1: A is stored [K][M] (lda).
//                                      b_layout 0: B is [K][N] (ldb); 1: B is stored [N][K] (ldb).
// split_k > 1: the K range is cut into split_k slabs, slab z writes C + z * c_split_stride (no bias/alpha fold
// is lost: alpha applied per slab, bias must be null) and the caller reduces with launch_splitk_reduce.
void launch_gemm(int a_layout, int b_layout, const float* A, const float* B, float* C, int M, int N, int K,
                 int lda, int ldb, int ldc, float alpha, const float* bias_n, int split_k, size_t c_split_stride,
                 hipStream_t s, double* colstats = nullptr,     // colstats [2][N] (no split-K): = Σ_rows C, Σ_rows C² (needs `sums`)
                 float* rowsq = nullptr, float rowsq_scale = 0.f, int* rowsq_parts = nullptr,
                 bool busy_chip = false,      // the launch lands next to long-running kernels of other streams (kernel choice)
                 const GridSumWs* sums = nullptr,       // workspace of the ordered column sums (required with colstats)
                 GemmSplitWs* split_ws = nullptr);      // planes of B for the split-bf16 kernel (null: exact-fp32 MFMA kernels)
// rowsq [gemm_rowsq_parts(N)][M]: rowsq_scale · Σ_cols C² per row, split by column tile (128 columns in the tiled kernel,
// 16 in the LDS-stationary one): *rowsq_parts = the number of parts this launch wrote; launch_sum_parts adds them in
// order (a runtime-length loop over the parts inside the row passes' unrolled gather was measured: it halves their
// speed, so the parts are combined once up front)
int gemm_rowsq_parts(int N);             // upper bound of *rowsq_parts: sizes the buffer
// LDS-stationary kernel for the batch-sized projection products (gemm_tstat.hip); false = shape not covered
bool launch_gemm_tstat(int a_layout, int b_layout, const float* A, const float* B, float* C, int M, int N, int K,
                       int lda, int ldb, int ldc, float alpha, const float* bias_n, hipStream_t s, double* colstats,
                       float* rowsq, float rowsq_scale, int* rowsq_parts, bool busy_chip = false, const GridSumWs* sums = nullptr);
// Row-panel kernel for per-rank batch sizes (gemm_rows.hip): a workgroup owns 32 rows and all N <= 320 columns.
// A [M][K] row-major; b_layout as launch_gemm. colstats (+ sums): ordered column sums of the output; rowsq [M]: COMPLETE
// rowsq_scale · Σ_cols C² per row (no parts); bn: the batch-norm backward of launch_bn_dx applied to the rows of A as they are
// loaded, dx written back over dy (b_layout 1 only); bn with pre == null: only grad_bias = (float) sums[k] (no batch-norm: what
// launch_colsum_finalize does). false: shape not covered, nothing launched.
bool launch_gemm_rows(int b_layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                      float alpha, const float* bias_n, hipStream_t s, double* colstats, const GridSumWs* sums, float* rowsq,
                      float rowsq_scale, const BnDxFused* bn);
bool gemm_rows_covers(int b_layout, int M, int N, int K, bool colstats, bool rowsq, bool bn);      // what launch_gemm_rows accepts
// The same decomposition on the bf16 matrix pipe (gemm_rsplit.hip; arithmetic of launch_gemm_split below): the workgroup's whole
// 32 x K panel of A cut into bf16 planes in LDS at once, B as planes in MFMA fragment order from L2 (GemmSplitWs::rplanes,
// gemm_rsplit_planes_bytes(N, K) bytes, cut here unless `rready`), a barrier-free K loop. Arguments as launch_gemm_rows.
size_t gemm_rsplit_planes_bytes(int N, int K);
void launch_gemm_rsplit_planes(int b_layout, const float* B, int N, int K, int ldb, void* planes, hipStream_t s);
// gf (forward product, b_layout 0, lda == K): the word gather-mean of launch_gather_mean rides on the staging of A — A is then
// WRITTEN (the phrase matrix [M][K], the gather kernel's bits) instead of read: see GatherFused.
bool launch_gemm_rsplit(int b_layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                        float alpha, const float* bias_n, hipStream_t s, double* colstats, const GridSumWs* sums, float* rowsq,
                        float rowsq_scale, GemmSplitWs* ws, const BnDxFused* bn, const GatherFused* gf = nullptr);
bool gemm_rsplit_covers(int b_layout, int M, int N, int K, bool colstats, bool rowsq, bool bn);
bool gemm_rsplit_gather_covers(int M, int N, int K, bool colstats, int window);      // ... with the gather inside
// Split-bf16 kernel for large batches (gemm_split.hip): fp32 operands cut exactly into three bf16 planes, nine (or six) bf16
// MFMAs per product with fp32 accumulation. rowsq [M]: COMPLETE rowsq_scale · Σ_cols C² per row. false: not covered / switched off.
// B travels as its three bf16 planes (GemmSplitWs::planes, gemm_split_planes_bytes(N, K) bytes, owned by the caller): cut by a
// small launch in front of the product unless `ready` says they are current — the caller clears `ready` whenever B changes.
size_t gemm_split_planes_bytes(int N, int K);
PlaneTarget gemm_split_plane_target(int N, int K, void* planes, int transposed);       // for TransformUpdateArgs::pt
PlaneTarget gemm_rsplit_plane_target(int N, int K, void* planes, int transposed);
void launch_gemm_split_planes(int b_layout, const float* B, int N, int K, int ldb, void* planes, hipStream_t s);
bool launch_gemm_split(int b_layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       float alpha, const float* bias_n, hipStream_t s, double* colstats, const GridSumWs* sums, float* rowsq,
                       float rowsq_scale, GemmSplitWs* ws, const BnDxFused* bn = nullptr);      // bn: as launch_gemm_rows (b_layout 1 only)
bool gemm_split_covers(int b_layout, int M, int N, int K, bool bn);      // shapes launch_gemm_split accepts
// dT = Aᵀ[M x rows] · B[rows x N], split-K over the rows (the batch), split-bf16 arithmetic (gemm_dt.hip): A [rows][M] (lda),
// B [rows][N] (ldb), partial [gemm_dt_slabs(rows, want)][M][N] — never more slabs than `want` —; the caller adds the slabs with
// launch_splitk_reduce. false: shape not covered, nothing launched.
bool gemm_dt_covers(int M, int N, int rows);
int gemm_dt_slabs(int rows, int want);
int gemm_dt_default_slabs(int rows, int cus);      // a slab per two CUs (two workgroups per slab), at least 64 rows each
bool launch_gemm_dt(const float* A, const float* B, float* partial, int M, int N, int rows, int lda, int ldb, int want_slabs, hipStream_t s);
// The same product in workgroups of ONE wave (gemm_dtw.hip, round 6): 32 x 64 tiles of the output per wave and slab, no LDS, no barrier —
// for per-rank batches, where the product runs next to both table passes and a whole-CU (or four-wave) workgroup waits for room.
bool gemm_dtw_covers(int M, int N, int rows);
int gemm_dtw_slabs(int rows, int want);
bool launch_gemm_dtw(const float* A, const float* B, float* partial, int M, int N, int rows, int lda, int ldb, int want_slabs, hipStream_t s);
int gemm_split_products();               // NVSM_GEMM_SPLIT: 6 (default), 9, or 0 = exact-fp32 MFMA kernels only
int gemm_rows_max_m();                   // largest M launch_gemm sends to the row-panel kernel (NVSM_GEMM_ROWS_MAX, default 8192; 0 = never): above it the split-bf16 kernel
float* gemm_dump_buffer();               // 256 B per device nobody reads (gemm_tstat.hip): the target of masked-out stores
void gemm_set_tstat_enabled(bool on);    // experiments / tests: force the tiled kernel
void launch_sum_parts(const float* parts, int nparts, int64_t stride, float* out, int64_t n, hipStream_t s);
void gemm_set_panel_enabled(bool on);    // experiments / tests: force the tiled kernel
int gemm_split_k_slabs(int K, int want);   // actual number of slabs launch_gemm will use for `want`
void launch_splitk_reduce(const float* partial, int slabs, size_t stride, float* out, int64_t n, hipStream_t s);

// ---- batch normalisation (F6, B5; replaces cuDNN per-activation BN, cpp/cudnn_utils.cu:82-183) --
// (the forward statistics μ, 1/sqrt(σ²+ε) are evaluated inside the loss kernel from the GEMM's column sums: LossArgs)
// dβ = Σdy, dγ = Σdy·x̂ (double sums [2][dim]) → floats, published together with grad_bias = dβ;
// dx = invσ·(dy − (dβ + x̂·dγ)/N), in place on dy
void launch_bn_dx(float* dy, const float* pre, const float* mean, const float* inv_std, const double* sums, float* dbeta,
                  float* dgamma, float* grad_bias, double n_global, int64_t rows, int dim, hipStream_t s);
void launch_colsum_finalize(const double* sums, int dim, float* out, hipStream_t s);   // no-BN grad_bias = Σdy

// codes a kernel stores into the engine's error word (page-locked host memory, read at the next synchronisation point)
enum { NVSM_BAD_WORD_ID = 1, NVSM_BAD_ENTITY_ID = 2, NVSM_SORT_TIMEOUT = 3, NVSM_NONFINITE_BASE = 16 };
// ids outside [0, limit) become row 0 and raise `code` in *err_flag (err_flag may be null: tests of single kernels)
void launch_narrow_i64(const int64_t* src, int* dst, int64_t n, int64_t limit, int* err_flag, int code, hipStream_t s);
// stamp[list[i]] = value for i < *count (launch_stamp_rows' job, riding on the prologue: list == null: none)
struct StampJob { const int* list; const int* count; int* stamp; int value; };
void launch_step_prologue(const int64_t* words64, int* widx, int64_t nW, const int64_t* labels, int64_t B, int R,
                          int64_t num_words, int64_t num_entities, uint64_t seed, uint64_t step, int* ids, double* stats,
                          int nstats, int* err_flag, hipStream_t s, StampJob stamps = StampJob{nullptr, nullptr, nullptr, 0});
void launch_check_finite(const float* x, int64_t n, int* err_flag, int code, hipStream_t s);   // NVSM_DEBUG (CHECK_MATRIX)
void launch_scale(float* x, int64_t n, float a, hipStream_t s);      // x *= a (replica averaging)
// Host batch → HBM by a kernel that READS page-locked host memory over PCIe (up to four arrays in one launch) instead of
// hipMemcpyAsync: the runtime's copy call held the calling thread for most of a step (0.55-0.8 ms per step measured for the
// four arrays of a 51 200-window batch while the GPU was busy), so the host never got ahead of the GPU and every step began
// with the GPU waiting for its launches. src: device-visible addresses of page-locked memory; bytes: multiples of 4.
struct HostPull { void* dst[4]; const void* src[4]; size_t bytes[4]; int count; };
void launch_host_pull(const HostPull& p, hipStream_t s);
// data parallel, one collective per step (model.cpp dp_fold_): tail = [gb (de floats) | loss as hi, lo floats] behind the projection
// gradient, and back (gather_gemm.hip)
void launch_dp_pack_tail(const float* gb, const double* loss, float* tail, int de, hipStream_t s);
void launch_dp_unpack_tail(const float* tail, float* gb, double* loss, int de, hipStream_t s);
void launch_delay(int microseconds, hipStream_t s);      // one wave spinning on the 100 MHz wall clock (profiling aid)
void launch_iota(int* dst, int64_t n, hipStream_t s);

// ---- fused loss forward + backward (F7–F16, B1–B4; replaces cpp/objective.cu:159-305,354-425 and the
// nonlinearity at cpp/params.cu:430-446,474-491) ------------------------------------------------
struct LossArgs {
    const float* pre;         // [B][de]  T·x (+b when !bn)
    const double* bn_sums;    // [2][de] (bn) Σx, Σx² over the batch, from the projection GEMM's epilogue
    double bn_n;              //      number of rows behind those sums (global batch with synchronised batch-norm)
    float bn_eps;
    float* bn_mean;           // [de] (bn) OUT: μ and 1/sqrt(σ²+ε), written by block 0 for the backward pass
    float* bn_inv_std;        // [de] (bn) OUT
    const float* bias;        // [de] (bn: β)
    const float* E;           // [nD][de]
    int64_t E_rows;           // nD (0: unknown) — which loop the gather runs depends on whether E fits the Infinity Cache
    LazyView lazyE;           // pending decay of E's rows, applied as they are gathered (stamp null: none)
    const int* ids;           // [B*R]
    const float* inst_w;      // [B] or null
    float* proj;              // [B][de]  act(BN(pre))
    float* dy;                // [B][de]  d/d(BN output) = gproj · act'
    float* coef;              // [B*R]   ±m_j  (signed multipliers)
    float* probs;             // [B*R]
    float* pp;                // [B]     mean_t(proj²)
    double* loss_acc;         // [1]     Σ ω·log p                           } written (not accumulated) by the ordered grid-wide sum;
    double* colstats;         // [2][de] Σdy, Σdy·x̂ (x̂ only when bn)      } colstats == loss_acc + 1
    GridSumWs sums;           // its workspace (kernels.h GridSumWs, one column group)
    int64_t B;
    int de, R, k;
    int bn, nonlinearity, rebalance;
    int l2_entity;            // l2_normalize_entity_reprs: gathered rows are divided by their norm (generic kernel only)
    float sig_eps, sig_hi, d_eps;
    double d_hi;              // 1 − d_eps compared in double (include/cuNVSM/cuda_utils.h:230)
    float inv_batch;          // exp(−log(B_global))
    float neg_scale;          // (k+1)/(2k)
    float clip_min, clip_max; // nextafter-widened hard_tanh bounds
    float inv_de;             // exp(−log(de))
};
void launch_loss(const LossArgs& a, hipStream_t s);
// true: the kernel launch_loss picks for these shapes applies LossArgs::lazyE itself; false: the rows must be current
bool loss_reads_lazily(int de, int R, bool l2_entity);
bool loss_two_row_sets(int64_t table_rows, int de);      // the row-gathering loss kernel keeps two sets of rows in flight per wave (tables beyond the Infinity Cache)

// per-row mean of squares: out[b] = Σ_t G[b][t]² · inv_dim  (cpp/updates_adam.cu:232-240)
void launch_row_meansq(const float* G, int64_t rows, int dim, float inv_dim, float* out, hipStream_t s);
// grad_entity[j][t] = coef[j] · proj[j/R][t]  (tests / gradient checker only)
void launch_materialize_grad_entity(const float* coef, const float* proj, int64_t N, int R, int de, float* out, hipStream_t s);
// ---- optional L2 row normaliser (cpp/cuda_utils.cu:12-130) ----
void launch_l2_rows_forward(const float* x, int64_t rows, int dim, float* y, float* norms, hipStream_t s);
void launch_l2_rows_backward(const float* g, const float* x, const float* norms, int64_t rows, int dim, float scale,
                             float* gin, float* msq, hipStream_t s);
void launch_materialize_grad_entity_l2(const float* coef, const float* proj, const float* E, const int* ids, int64_t N, int R,
                                       int de, float* out, float* msq, hipStream_t s);

// ---- batch → CSR by table row (replaces the atomic scatter of update_repr_kernel, cpp/storage.cu:37-49)
// Stable radix sort of (key, value) pairs by the low `bits` bits of the key (sort.hip). `temp` must be zero-filled ONCE when
// allocated; *epoch (host, starts at 0, one per workspace) is the value its arrival counter has reached. vals_in may be
// null (= 0, 1, 2, …). err_flag: the engine's error word (a grid-wide wait that never completes stores NVSM_SORT_TIMEOUT).
// zero_buf (optional, 16 B aligned, zero_count % 4 == 0): ints the first pass clears on the way (the CSR's per-step counters).
size_t sort_pairs_temp_bytes(int64_t n, int bits);
// gate (optional, device): the launches return at once when *gate == 0 — the outputs are then left as they were (the chunk
// order of a table none of whose rows has chunks this step).
void sort_pairs(void* temp, size_t temp_bytes, uint64_t* epoch, const int* keys_in, int* keys_out, const int* vals_in,
                int* vals_out, int64_t n, int bits, int* err_flag, hipStream_t s, int* zero_buf = nullptr, int64_t zero_count = 0,
                const int* gate = nullptr);
struct Csr {
    int* sorted_key;      // [n]
    int* sorted_entry;    // [n]  entry ids ordered by row (stable)
    int* row_begin;       // [rows]
    int* row_end;         // [rows]
    int* chunk_base;      // [rows]  first level-1 chunk of a long row
    int* chunk_desc;      // [max_chunks][3] row, begin, end
    int* chunk2_base;     // [rows]  first level-2 chunk of a very long row
    int* chunk2_desc;     // [max_chunks2][2] first, last (exclusive) level-1 chunk
    int* chunk_order;     // [max_chunks] level-1 chunks ordered by the position of their first entry in the batch (null: as numbered)
    int* num_chunks;      // [2]  level-1 / level-2 chunks in use
    int* num_touched;     // [1]  rows with at least one entry (num_chunks + 2)
    int* touched;         // [min(rows, max entries)] those rows, in no particular order (rows are independent)
    float* partial;       // [max_chunks][dim]
    float* partial_q;     // [max_chunks]
    float* partial2;      // [max_chunks2][dim]
    float* partial2_q;    // [max_chunks2]
    int* arrive_row;      // [rows]        arrival counters of table_pass_kernel, zero between launches
    int* arrive2;         // [max_chunks2]
    int64_t n;            // entries
    int64_t rows;
    int max_chunks, max_chunks2;
    int chunk;            // entries per level-1 chunk of a long row this step: kChunk, or kChunkSmall (Model::chunk_entries)
};
constexpr int kChunk = 64;    // entries per level-1 chunk of a long row
constexpr int kChunkSmall = 32;       // (48 / 20 / 16: LSE 0.1533 / 0.1506 / 0.1534 ms against 0.1503; 64: 0.1568)      // ... for the small batches of SGD / Adagrad handles (Model::chunk_entries)
constexpr int64_t kChunkSmallMaxEntries = 131072;
constexpr int kFan = 32;      // level-1 partials per level-2 chunk (64 until round 5: a row of 33-64 chunks then ended in one sum of up to 64 partials;
                              //   32 / 128: batch 51 200 and 6 400 the same / +0.4 %, +1.2 %, LSE batch 4 096 0.1473 -> 0.1456 ms / the same)

// bounds + long-row chunk list, from sorted_key; counters_cleared: row_begin | row_end | num_chunks | num_touched (one
// allocation of csr_counter_ints(rows) ints) are already zero (sort_pairs cleared them), otherwise a memset does it
// order_key (optional, [max_chunks], with Csr::chunk_order): the chunk fill also writes the keys of launch_chunk_order (keys_written)
void launch_csr_build(const Csr& c, hipStream_t s, bool counters_cleared = false, int* order_key = nullptr);
// Csr::chunk_order: the level-1 chunks by where in the batch their entries start (512 buckets; update.hip table_pass_kernel
// gives each XCD one eighth of the batch so that the chunks of different hot rows over the same windows share an L2).
// key_in / key_out: [max_chunks] scratch; sort_temp: a sort_pairs workspace for max_chunks pairs
void launch_chunk_order(const Csr& c, int* key_in, int* key_out, void* sort_temp, size_t sort_temp_bytes, hipStream_t s, bool keys_written = false);
inline int64_t csr_counter_ints(int64_t rows) { return (2 * rows + 3 + 63) / 64 * 64; }
bool row_pass_split(const Csr& c);                    // rows · ratio >= entries: touched-row list + streaming pass over the rest
double table_split_ratio();                           // that ratio (2 unless NVSM_SPLIT_RATIO says otherwise)

// ---- row passes: gather Σ coef·X[src] per table row, then the optimiser's row-local formula --------
enum RowKind {
    ROW_SGD = 0,               // P = P·decay + lr·g                         (cpp/storage.cu:51-102)
    ROW_ADAGRAD_ENT = 1,       // a += q; P = P·decay + lr·g/sqrt(a+ε)       (cpp/updates_adagrad.cu:99-179, window 1)
    ROW_ADAM_MV = 2,           // m = β1 m + (1−β1) g; v = β2 v + (1−β2) q   (cpp/updates_adam.cu:196-252)
    ROW_ADAM_SPARSE_ENT = 3,   // ROW_ADAM_MV then P = P·decay + lr·cnt·bc·m/(√v+ε)   (:332-384, window 1)
    ROW_ADAM_DENSE = 4,        // ROW_ADAM_MV then P = P·decay + lr·bc·m/(√v+ε)       (:293-311)
    ROW_ADAM_FULL = 5,         // m,v with L2 folded in, v per element; P += lr·bc·m/(√v+ε)  (:203-213,253-282,312-328)
    ROW_SCALAR_ACC = 6         // a += q only (Adagrad, window > 1)           (cpp/updates_adagrad.cu:136-158)
};
struct RowPassArgs {
    int table;                 // 0 = words (entry e: src = e / div, coef = wts[e]), 1 = entities (src = j / div, coef = coefs[j])
    int kind;
    const float* X;            // [num_src][dim]   gradient source rows
    const float* wts;          // words: per-entry weights or null
    const float* coefs;        // entities: signed multipliers
    const float* sq_src;       // words: msq[src]; entities: pp[src]
    const float* src_scale;    // optional per-src scale (Adagrad, window > 1)
    uint32_t div;              // window (words) or R (entities)
    uint64_t div_magic;        // floor(2^37 / div) + 1: src = (entry * div_magic) >> 37, exact for entry < 2^26, div <= 2048
    float* P; float* m; float* v;   // table, first moment [rows][dim], per-element second moment (ROW_ADAM_FULL only)
    const float* sc_in;        // per-row scalar state in  (Adam v / Adagrad a, [rows]); ping-pong with sc_out because
    float* sc_out;             //   every lane of a row's thread group reads the old value
    int dim;
    float lr, decay, lambda;
    float one_m_b1, s_m, one_m_b2, s_v, bc, eps, c_reg;
    int dense;                 // visit rows without entries (decay / dense Adam)
    int max_blocks;            // 0 = one thread group per row; > 0 = cap the grid (rows are grid-strided) so that a kernel
                               //   running concurrently on another stream finds free registers on every CU
    int touched_only;          // set by launch_row_pass: visit the rows of Csr::touched only (see row_pass_split)
    int nt_p;                  // the table rows themselves likewise (experiments)
    int nt_m;                  // first moments with streaming (nt) loads / stores: nobody gathers them (documents table)
    int shallow;               // set by launch_row_pass: two entries in flight per lane instead of eight (rows >= entries)
    int lazy;                  // lazy dense decay (below): rows without entries are NOT visited, their decay stays pending
    int rows_elsewhere;        // set by launch_table_pass: the rows of at most a chunk's entries are done by entry_walk_kernel
    int wide;                  // this ROW_SGD pass belongs to an Adam update (update.hip: the WIDE family of the one-launch pass)
    int untouched_done;        // the streaming pass over the rows WITHOUT entries of a split dense pass has been queued already (launch_untouched_rows)
    LazyView pending;          // lazy: the row's P (and m, by s_m) first get the factors of the updates (stamp[row], now] the
                               //   row sat out, one by one (pending.stamp null: the rows are current)
};
// ---- lazy dense decay, for tables much larger than the batch ------------------------------------------------------------
// The reference rewrites EVERY row of a table on every update (θ·(1 − λ·lr), Adam m·β₁, v·β₂: cpp/storage.cu:65-67,
// cpp/updates_adam.cu:196-252): at |D| = 2 M that is 8 of the 10 GB a step moves. A row without entries only ever gets
// multiplied by those per-update constants, so the multiplications can wait until somebody looks at the row: every row
// carries the number of updates applied to it (stamp), and whoever reads a row applies the factors of the updates it sat
// out ONE BY ONE, in fp32, in update order (the factors of the last kLazyHistory updates travel in the kernel arguments:
// LazyView) — exactly the sequence of roundings the dense passes would have produced, so parameters and optimiser state
// stay bit-identical to the eager path (tests/test_gpu_lazy.py compares them):
//   * the gathers of the forward pass (word gather-mean, loss kernel) do it in registers and write nothing back;
//   * the row passes of the update do it to the rows of the batch before the optimiser's formula and store the result;
//     the per-row scalar (Adam v / Adagrad accumulator) of those rows is brought up to date into a snapshot buffer by a
//     small launch in front of the passes (launch_lazy_refresh, scalars_only), and the rows' stamps are set by a small
//     launch behind the last pass (launch_stamp_rows) — not inside it, where another wave of the row's thread group could
//     still be about to read the old stamp;
//   * every kLazyHistory updates, and before anything else looks at a whole table (get_param, set_param, replica
//     averaging), launch_lazy_refresh brings all rows up to date.
struct LazyRefreshArgs {
    float* P; float* m;                 // table rows, first moments (null: none)
    float* sc;                          // per-row scalar state (Adam v / Adagrad accumulator; null: none)
    float* sc_snapshot;                 // copy of the refreshed scalar the row pass reads while it writes `sc` (null: none)
    int scalars_only;                   // P, m and the stamps are left alone (the row pass refreshes them itself): only the
                                        //   scalar is brought up to date and snapshotted, one thread per row
    int* stamp;                         // [rows] updates applied to the row
    const int* list; const int* list_count;      // rows to refresh (Csr::touched); null = all `rows`
    int64_t rows; int dim;
    int now;                            // updates applied to the table so far
    float s_m, s_v;                     // per-update factors of m and of the scalar (1 = none)
    float decay[kLazyHistory];          // factor of update u on P at [(u - 1) % kLazyHistory]
};
// stamp[row] = value for the rows of c.touched (after the last pass of a lazy table's update)
void launch_stamp_rows(const Csr& c, int* stamp, int value, int64_t max_rows, hipStream_t s);
void launch_lazy_refresh(const LazyRefreshArgs& a, int64_t max_rows, hipStream_t s);

void launch_chunk_pass(const Csr& c, const RowPassArgs& a, hipStream_t s);
// untouched_s: the stream the streaming pass over the rows WITHOUT entries of a split dense pass is queued on (null: `s`).
// Those rows and the rows with entries are disjoint, so the two launches need no order between them — only the CSR bounds
// in front of both and the table's next reader behind both.
void launch_row_pass(const Csr& c, const RowPassArgs& a, hipStream_t s, hipStream_t untouched_s = nullptr);
// The streaming half of a split dense pass on its own: the rows WITHOUT entries of a table much larger than the batch get the
// row formula with g = 0 (the decay). It needs the CSR's row bounds and nothing else of the step, so the fused step queues it
// right behind the CSR build, under the forward pass, instead of in the update's tail; the pass proper is then launched with
// RowPassArgs::untouched_done. true: launched (the pass is split, dense, not lazy, its kind row-local for untouched rows).
bool launch_untouched_rows(const Csr& c, const RowPassArgs& a, hipStream_t s);
// both, in one launch (update.hip). Returns how the rows with entries were walked (tests assert which path a shape took).
enum TablePassPath { TABLE_PASS_DENSE = 0, TABLE_PASS_LIST_WALK = 1, TABLE_PASS_ENTRY_WALK = 2 };
int launch_table_pass(const Csr& c, const RowPassArgs& a, hipStream_t s, hipStream_t untouched_s = nullptr);
void set_table_pass_one_launch(bool on);      // tests / A-B runs: false = the three-launch form (also NVSM_MERGED_PASS=0)

// words, window > 1 (cpp/updates_adagrad.cu:83-97, cpp/updates_adam.cu:132-151)
void launch_adagrad_scale(const float* acc, const int* idx, int window, int64_t B, float eps, float* scale, hipStream_t s);
void launch_adam_u(const float* m, const float* v, int dim, const int* idx, int window, int64_t B, float bc, float eps,
                   float* U, hipStream_t s);

// ---- dense projection optimiser (U4; cpp/updates.cu:24-34, updates_adagrad.cu:33-70, updates_adam.cu:46-105)
struct TransformUpdateArgs {
    PlaneTarget pt[2];
    int de;                          // entity dimension: T is stored [dw][de]
    // the split-K slabs of the dT product, added up here instead of by launch_splitk_reduce in front of the update (null: gT holds
    // the gradient): the same sums in the same order — 16 interleaved groups of slabs, then the groups in order — one launch less
    const float* partial; int slabs; size_t slab_stride;
    float* T; float* b;              // parameters (nT = de*dw, nb = de)
    float* gT; float* gb;            // gradients (overwritten with the applied direction, as the reference does)
    float* s0T; float* s0b; float* s1T; float* s1b;
    int nT, nb, method;
    float lr, lambda, eps, one_m_b1, s_m, one_m_b2, s_v, bc;
};
void launch_transform_update(const TransformUpdateArgs& a, hipStream_t s);


}  // namespace cunvsm
