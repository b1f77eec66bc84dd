// fp32 MFMA GEMM for the two batch-sized projection products at PER-RANK batch sizes (a few thousand to ~16 k rows: the
// 6 400 windows a rank of the 8-GPU job holds, the LSE recipe's 4 096):
//   forward   pre[B][d_e]     = phrase[B][d_w] · Tt[d_w][d_e]  (+ bias, + batch-norm column sums)   cpp/params.cu:417
//   backward  gphrase[B][d_w] = alpha · dx[B][d_e] · T          (+ per-row mean of squares)          cpp/objective.cu:453
//             with the batch-norm backward dx = invσ·(dy − (dβ + x̂·dγ)/N) (cpp/cudnn_utils.cu:143-183) applied to the
//             rows of dy as they are loaded (and written back: the dT product reads dx afterwards)
// At these sizes neither of the large-batch kernels fits: the 128 x 128 tiled kernel has 100 workgroups for 256 CUs, the
// LDS-stationary kernel (gemm_tstat.hip) spends its time filling 160 KB of LDS per workgroup for three row blocks each and
// cannot start on a CU while a sort workgroup holds LDS there (36 / 30-155 us for 0.98 GFLOP at B = 6 400).
// Here a workgroup owns 32 rows and ALL columns of the output (200 workgroups at B = 6 400, one per CU, 50 KB of LDS):
//   * every element of A is read exactly once by exactly one workgroup — which is what lets the batch-norm backward ride on
//     the load (one launch instead of bn_dx + GEMM) and the rows' sums of squares come out complete (no per-tile parts
//     and no launch to add them up);
//   * wave w multiplies the 32 rows with its own two 32-column tiles (v_mfma_f32_32x32x2_f32, exact fp32; 4 waves for
//     N = 256, 5 for N = 300), K in tiles of 32 staged through LDS with the next tile's global loads in flight;
//   * the projection matrix (307 KB) is re-read per workgroup from L2, 61 MB in all: a few microseconds of L2 bandwidth.
// Sums are formed in a fixed order: results are the same bits every run.
#include "kernels.h"
#include "device_utils.h"

#include <atomic>
#include <cstdlib>

namespace cunvsm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RM = 32, RK = 32;
constexpr size_t kRowsLdsBytes = 128 * 1024;   // two A + two B images; more than 64 KB needs the per-kernel, per-device opt-in
constexpr int kRowsMaxDevices = 64;

struct RowsArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    float alpha;
    const float* bias_n;
    double* colstats; GridSumWs sums;          // forward: [2][N] = Σ_rows C, Σ_rows C² (ordered grid-wide sum)
    float* rowsq; float rowsq_scale;           // backward: rowsq[M] = rowsq_scale · Σ_cols C²
    // PRE: batch-norm backward on the rows of A (= dy, overwritten with dx)
    float* A_rw; const float* pre; const float* mean; const float* inv_std; const double* bn_sums;
    float* dbeta; float* dgamma; float* grad_bias; float inv_n;
    float* dump;                               // 256 B nobody reads: where masked-out global stores go
#ifdef NVSM_ROWS_DBG
    int dbg;                                   // experiments: 1 = no multiply, 2 = no tile loads in the loop, 4 = no LDS stores, 8 = no epilogue
#endif
};
#ifdef NVSM_ROWS_DBG
#define ROWS_DBG(g, bit) ((g).dbg & (bit))
#else
#define ROWS_DBG(g, bit) 0
#endif

// BLAY 0: B is [K][N] (ldb), LDS image Bs[k][NP]; BLAY 1: B stored [N][K] (ldb), LDS image Bs[n][RK + 1].
// TPW: 32-column tiles per wave (2; 1 for N <= 128, so that four waves still share the work).
// WAVES: 4, or 5 for 256 < N <= 320 (a register budget of 512 per lane with four waves, 256 with five).
//
// Pipeline. One workgroup per CU means one wave per SIMD: nothing hides a load but the MFMAs of the same wave, and a tile's
// MFMAs (0.85 us) are shorter than a load's way from L2 / HBM (1.5-2 us under load). So: two register stages, each loaded TWO
// tiles ahead of its use; two LDS images, tile t + 1 written while tile t is multiplied (one barrier per tile); and every
// global load, global store and LDS store of the loop body unconditional — lanes / tiles outside the matrix load a clamped
// address, have their values replaced by zero when they are stored, and store to a dump address — because the waits for
// the prefetched tiles are counted (s_waitcnt vmcnt(n)) and the compiler can only count what is issued on every path: the
// first version, with its loads under bounds branches, waited for everything in flight at every step and ran 4-6 us per tile.
template <int BLAY, bool PRE, int TPW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_rows_kernel(RowsArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int nthreads = 64 * WAVES, waves = WAVES;
    constexpr int NB = 4 * TPW;                                // float4 of the B tile per thread (RK N / 4 over 64 WAVES threads)
    constexpr int WN = 32 * TPW;                               // columns per wave
    constexpr int NP = WAVES * WN;                             // columns covered by the waves' tiles (>= N)
    constexpr int AIMG = RM * (RK + 1) + 8;                    // floats per A image (padded to 16 B)
    constexpr int BIMG = (BLAY == 0) ? RK * NP : ((NP * (RK + 1) + 3) & ~3);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.x * RM;
    float* As0 = lds;
    float* As1 = lds + AIMG;
    float* Bs0 = lds + 2 * AIMG;
    float* Bs1 = Bs0 + BIMG;
    float* sink = Bs1 + BIMG;                                  // 8 floats nobody reads: LDS stores of lanes without a slot
    float* consts = sink + 8;                                  // PRE: [4][K] μ, invσ, dβ, dγ
    int* flag = reinterpret_cast<int*>(consts + (PRE ? 4 * g.K : 0));
    float* red = reinterpret_cast<float*>(flag + 4);           // epilogue scratch: [waves][RM] or [2][NP]

    if (PRE) {
        for (int k = tid; k < g.K; k += nthreads) {
            consts[k] = g.mean[k]; consts[g.K + k] = g.inv_std[k];
            const float db = static_cast<float>(g.bn_sums[k]), dg = static_cast<float>(g.bn_sums[g.K + k]);      // cudnn_utils.cu:158-173
            consts[2 * g.K + k] = db; consts[3 * g.K + k] = dg;
            if (blockIdx.x == 0) { g.dbeta[k] = db; g.dgamma[k] = dg; g.grad_bias[k] = db; }      // ∂β is the bias gradient; ∂γ is dropped (:173)
        }
    }
    if (!PRE && g.grad_bias && blockIdx.x == 0)      // no batch-norm: the bias gradient is Σdy (launch_colsum_finalize's job, riding here)
        for (int k = tid; k < g.K; k += nthreads) g.grad_bias[k] = static_cast<float>(g.bn_sums[k]);
    // the B images' columns N .. NP are never written by store_tile: zero them once (they feed MFMAs whose results are dropped,
    // but NaN bit patterns left in LDS would poison the row sums of squares otherwise)
    for (int i = tid; i < 2 * BIMG; i += nthreads) Bs0[i] = 0.f;
    __syncthreads();

    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // A: RM x RK = 256 float4, one per thread (threads >= 256): row tid >> 3, k offset 4 (tid & 7)
    // B: BLAY 0: RK rows x N/4 float4; BLAY 1: N rows x RK/4 float4 — strided over the threads; where a thread's float4s sit in
    // the tile (global offset, LDS offset, k inside the tile) is worked out once, not per tile (a division by N / 4 each).
    const int a_row = (tid >> 3) & (RM - 1), a_kq = (tid & 7) << 2;
    const bool a_mine = tid < 256;
    const bool a_row_ok = a_mine && m0 + a_row < g.M;
    const size_t a_off = static_cast<size_t>(a_row_ok ? m0 + a_row : 0) * g.lda;      // (row start: the k offset is added per tile, clamped)
    const int a_loff = a_mine ? a_row * (RK + 1) + a_kq : -1;
    const int nb_f4 = (BLAY == 0) ? RK * (g.N >> 2) : g.N * (RK >> 2);
    int b_goff[NB], b_loff[NB], b_k[NB];      // b_loff < 0: the slot is not this thread's
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        const int f = tid + it * nthreads;
        b_goff[it] = 0; b_loff[it] = -1; b_k[it] = 0;
        if (f < nb_f4) {
            if (BLAY == 0) {
                const int kk = f / (g.N >> 2), n4 = f - kk * (g.N >> 2);
                b_goff[it] = kk * g.ldb + 4 * n4; b_loff[it] = kk * NP + 4 * n4; b_k[it] = kk;
            } else {
                const int n = f >> 3, kq = (f & 7) << 2;
                b_goff[it] = n * g.ldb + kq; b_loff[it] = n * (RK + 1) + kq; b_k[it] = kq;
            }
        }
    }
    const int ntiles = (g.K + RK - 1) / RK;
    struct Stage { float4 a, x, b[NB]; };
    // tile `t` (clamped to the last one: the loop body loads up to three tiles past the end) into a register stage
    auto load_tile = [&](int t, Stage& st) {
        if (ROWS_DBG(g, 2) && t > 1) return;
        const int k0 = min(t, ntiles - 1) * RK;
        const int ka = (k0 + a_kq < g.K) ? k0 + a_kq : 0;                  // K % 4 == 0: a float4 is in or out as a whole
        st.a = *reinterpret_cast<const float4*>(g.A + a_off + ka);
        if (PRE) st.x = *reinterpret_cast<const float4*>(g.pre + a_off + ka);
        const float* bt = g.B + ((BLAY == 0) ? static_cast<size_t>(k0) * g.ldb : static_cast<size_t>(k0));
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int off = (k0 + b_k[it] < g.K) ? b_goff[it] : ((BLAY == 0) ? b_goff[it] - b_k[it] * g.ldb : b_goff[it] - b_k[it]);
            st.b[it] = *reinterpret_cast<const float4*>(bt + off);
        }
    };
    // a register stage that holds tile `t` into an LDS image (values outside the matrix become zero here)
    auto store_tile = [&](int t, Stage& st, float* As, float* Bs) {
        if (ROWS_DBG(g, 4)) return;
        const int k0 = t * RK;
        const bool a_ok = a_row_ok && k0 + a_kq < g.K;
        float v[4] = {st.a.x, st.a.y, st.a.z, st.a.w};
        if (PRE) {
            // dx = invσ · (dy − (dβ + x̂·dγ) / N), x̂ = (x − μ)·invσ      (bn_dx_kernel, loss_bn.hip); written back over dy
            const float x[4] = {st.x.x, st.x.y, st.x.z, st.x.w};
            const int k = a_ok ? k0 + a_kq : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float is = consts[g.K + k + i];
                const float xhat = (x[i] - consts[k + i]) * is;
                v[i] = is * (v[i] - (consts[2 * g.K + k + i] + xhat * consts[3 * g.K + k + i]) * g.inv_n);
            }
            *reinterpret_cast<float4*>(a_ok ? g.A_rw + a_off + k0 + a_kq : g.dump) = make_float4(v[0], v[1], v[2], v[3]);
        }
        float* ap = (a_loff >= 0) ? As + a_loff : sink;
#pragma unroll
        for (int i = 0; i < 4; ++i) ap[i] = a_ok ? v[i] : 0.f;
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const bool ok = k0 + b_k[it] < g.K;
            const float4 b = ok ? st.b[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            float* bp = (b_loff[it] >= 0) ? Bs + b_loff[it] : sink;
            if (BLAY == 0) *reinterpret_cast<float4*>(bp) = b;
            else { bp[0] = b.x; bp[1] = b.y; bp[2] = b.z; bp[3] = b.w; }
        }
    };
    auto multiply_tile = [&](const float* As, const float* Bs) {
        if (ROWS_DBG(g, 1)) return;
        float af[RK / 2], bf[RK / 2][TPW];
#pragma unroll
        for (int s = 0; s < RK / 2; ++s) {
            const int k = 2 * s + lk;
            af[s] = As[l31 * (RK + 1) + k];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int n = wid * WN + j * 32 + l31;
                bf[s][j] = (BLAY == 0) ? Bs[k * NP + n] : Bs[n * (RK + 1) + k];
            }
        }
#pragma unroll
        for (int s = 0; s < RK / 2; ++s)
#pragma unroll
            for (int j = 0; j < TPW; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[s][j], af[s], acc[j], 0, 0, 0);
    };

    Stage s0, s1;
    load_tile(0, s0);
    load_tile(1, s1);
    store_tile(0, s0, As0, Bs0);
    load_tile(2, s0);
    __syncthreads();
    for (int t = 0; t < ntiles; t += 2) {
        // even tile t lives in image 0; tile t + 1 (stage 1) goes into image 1 while it is multiplied
        store_tile(t + 1, s1, As1, Bs1);
        load_tile(t + 3, s1);
        multiply_tile(As0, Bs0);
        __syncthreads();
        if (t + 1 >= ntiles) break;
        store_tile(t + 2, s0, As0, Bs0);
        load_tile(t + 4, s0);
        multiply_tile(As1, Bs1);
        __syncthreads();
    }

    if (ROWS_DBG(g, 8)) { if (acc[0][0] == 123.456f) g.C[0] = 1.f; return; }
    // The MFMA operands were fed swapped (the tile is computed transposed): a lane owns 4 x 4 consecutive columns of ONE row,
    // acc[j][4 q + t] = C[m0 + l31][wid WN + j 32 + 8 q + 4 lk + t] — 16 B stores (one column of 16 rows per lane, the natural
    // layout, is 32 scalar stores per lane: 12 of the forward product's 35 us) and row sums of squares by one cross-lane add.
    // Column sums go through LDS: the images are dead behind the loop's last barrier, every lane drops its values into a
    // [32][NP + 4] tile and thread c adds column c over the 32 rows in row order.
    const int row = m0 + l31;
    constexpr int TP = NP + 4;
    float* tile = Bs0;                                          // 32 (NP + 4) floats <= the two B images
    float rsq = 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wid * WN + j * 32 + 8 * q + 4 * lk;
            const bool ok = row < g.M && col < g.N;                     // N % 4 == 0: the lane's four columns are in or out together
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = ok ? g.alpha * acc[j][4 * q + t] + (g.bias_n ? g.bias_n[col + t] : 0.f) : 0.f;
            if (ok) *reinterpret_cast<float4*>(g.C + static_cast<size_t>(row) * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t) rsq += v[t] * v[t];
            if (g.colstats) *reinterpret_cast<float4*>(tile + l31 * TP + col) = make_float4(v[0], v[1], v[2], v[3]);
        }
    if (g.rowsq) {
        rsq += __shfl_xor(rsq, 32);
        if (lk == 0) red[wid * RM + l31] = rsq;
        __syncthreads();
        if (tid < RM && m0 + tid < g.M) {
            float s = 0.f;
            for (int w = 0; w < waves; ++w) s += red[w * RM + tid];       // wave order: the same sum every run
            g.rowsq[m0 + tid] = s * g.rowsq_scale;
        }
    }
    if (g.colstats) {
        __syncthreads();
        for (int c = tid; c < NP; c += nthreads) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
            for (int r = 0; r < RM; ++r) { const float v = tile[r * TP + c]; s1 += v; s2 += v * v; }      // (rows >= M hold zeros)
            red[c] = s1; red[NP + c] = s2;
        }
        __syncthreads();
        const int N = g.N;
        double* out = g.colstats;
        grid_sum_ordered<nthreads>(g.sums.part, g.sums.part2, g.sums.arrive, g.sums.fan, 2 * N, static_cast<int>(blockIdx.x),
                                   static_cast<int>(gridDim.x), [&](int i) -> float { return i < N ? red[i] : red[NP + (i - N)]; },
                                   [&](int i, double v) { out[i] = v; }, flag);
    }
}

template <int BLAY, bool PRE, int TPW, int WAVES>
static bool rows_launch(const RowsArgs& g, int grid, size_t lds, hipStream_t s) {
    // more than 64 KB of dynamic LDS is an opt-in per kernel and per DEVICE (see tstat_launch_epi)
    static std::atomic<bool> attr_set[kRowsMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kRowsMaxDevices) return false;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_kernel<BLAY, PRE, TPW, WAVES>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kRowsLdsBytes)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev].store(true, std::memory_order_release);
    }
    NVSM_LAUNCH((gemm_rows_kernel<BLAY, PRE, TPW, WAVES>), dim3(grid), dim3(64 * WAVES), lds, s, g);
    return true;
}

// LDS bytes of a launch, 0 when the shape is not covered: N <= 320, N and K multiples of 4, everything in 64 KB of LDS
static size_t rows_plan(int b_layout, int M, int N, int K, bool colstats, bool rowsq, bool bn, int* tpw_out, int* waves_out) {
    if (M <= 0 || N <= 0 || K <= 0 || N > 320 || (N % 4) || (K % 4)) return 0;
    if ((colstats && rowsq) || (b_layout == 0 && (bn || rowsq)) || (bn && (b_layout != 1 || colstats))) return 0;
    if (b_layout == 0 && N > 256) return 0;          // (B as [K][N] with more than 256 columns: no caller)
    // One 32-column tile per wave and two waves per SIMD (one wave's loads, LDS stores and epilogue run under the other's
    // MFMAs), or two tiles per wave and one wave per SIMD (half the LDS fragment reads per MFMA). Alone, forward / backward at
    // 6 400 rows: 29.0 / 24.4 us against 34.7 / 29.7 us; step at batch 6 400 0.314 against 0.329 ms (interleaved A/B).
    // NVSM_ROWS_TPW=2: the two-tile form.
    const int tpw_env = tuning().rows_tpw;
    const int tpw = (N > 128 && tpw_env == 2) ? 2 : 1;
    const int wn = 32 * tpw;
    // (>= 256 threads: the A tile is one float4 per thread; rounded up to an instantiated workgroup size)
    const int waves = tpw == 2 ? (N <= 256 ? 4 : 5) : (N <= 128 ? 4 : (N <= 256 ? 8 : 10));
    const int threads = waves * 64, NP = waves * wn;
    const int nb_f4 = (b_layout == 0) ? RK * (N / 4) : N * (RK / 4);
    if ((nb_f4 + threads - 1) / threads > 4 * tpw) return 0;
    const size_t b_img = (b_layout == 0) ? static_cast<size_t>(RK) * NP : ((static_cast<size_t>(NP) * (RK + 1) + 3) & ~size_t(3));
    const size_t scratch = static_cast<size_t>(colstats ? 2 * NP : waves * RM);
    const size_t lds = (2 * (RM * (RK + 1) + 8) + 2 * b_img + 8 + (bn ? 4 * static_cast<size_t>(K) : 0) + 4 + scratch) * sizeof(float);
    if (lds > kRowsLdsBytes) return 0;
    *tpw_out = tpw; *waves_out = waves;
    return lds;
}
bool gemm_rows_covers(int b_layout, int M, int N, int K, bool colstats, bool rowsq, bool bn) {
    int t, w;
    return rows_plan(b_layout, M, N, K, colstats, rowsq, bn, &t, &w) != 0;
}

// true: launched. A [M][K] row-major, 16 B aligned operands, leading dimensions multiples of 4.
bool launch_gemm_rows(int b_layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                      float alpha, const float* bias_n, hipStream_t s, double* colstats, const GridSumWs* sums, float* rowsq,
                      float rowsq_scale, const BnDxFused* bn) {
    if ((lda % 4) || (ldb % 4) || (ldc % 4)) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) % 16) return false;
    int tpw = 0, waves = 0;
    const bool fused_bn = bn && bn->pre;
    const size_t lds = rows_plan(b_layout, M, N, K, colstats != nullptr, rowsq != nullptr, fused_bn, &tpw, &waves);
    if (!lds) return false;
    const int grid = (M + RM - 1) / RM;
    RowsArgs g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.bias_n = bias_n;
    g.colstats = colstats; g.rowsq = rowsq; g.rowsq_scale = rowsq_scale;
    if (colstats) {
        const int fan = grid_sum_fan(grid);
        if (!sums || sums->contrib_cap < grid || sums->width_cap < 2 * N || sums->groups_cap < (grid + fan - 1) / fan) return false;
        g.sums = *sums; g.sums.fan = fan;
    }
    if (bn && !bn->pre) {      // bias gradient only (no batch-norm)
        g.bn_sums = bn->sums; g.grad_bias = bn->grad_bias;
    } else if (bn) {
        g.A_rw = bn->dy; g.pre = bn->pre; g.mean = bn->mean; g.inv_std = bn->inv_std; g.bn_sums = bn->sums;
        g.dbeta = bn->dbeta; g.dgamma = bn->dgamma; g.grad_bias = bn->grad_bias; g.inv_n = static_cast<float>(1.0 / bn->n_global);
    }
    g.dump = gemm_dump_buffer();
    if (!g.dump) return false;
#ifdef NVSM_ROWS_DBG
    g.dbg = tuning().rows_dbg;
#endif
#define NVSM_ROWS_CASE(T, W)                                                                          \
    if (tpw == T && waves == W) {                                                                     \
        if (b_layout == 0) { if constexpr (W == 4 || W == 8) return rows_launch<0, false, T, W>(g, grid, lds, s); else return false; } \
        if (fused_bn) return rows_launch<1, true, T, W>(g, grid, lds, s);                                   \
        return rows_launch<1, false, T, W>(g, grid, lds, s);                                          \
    }
    NVSM_ROWS_CASE(1, 4) NVSM_ROWS_CASE(2, 4) NVSM_ROWS_CASE(2, 5) NVSM_ROWS_CASE(1, 8) NVSM_ROWS_CASE(1, 10)
#undef NVSM_ROWS_CASE
    return false;
}

int gemm_rows_max_m() {
    return tuning().gemm_rows_max;
}

}  // namespace cunvsm
