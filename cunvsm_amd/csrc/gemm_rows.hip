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

#include <cstdlib>

namespace cunvsm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RM = 32, RK = 32;
constexpr int kRowsMaxWaves = 5;               // N <= 320
constexpr int kRowsMaxBF4 = 10;                // float4 of the B tile per thread: 8 N / threads, threads = 32 N' (N' = N in 64s)

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
};

// BLAY 0: B is [K][N] (ldb), LDS image Bs[k][NP]; BLAY 1: B stored [N][K] (ldb), LDS image Bs[n][RK + 1].
// SWAP: operands fed swapped (the tile is computed transposed): a lane owns 4 x 4 consecutive columns of ONE row — 16 B
// stores and row sums of squares; !SWAP: a lane owns one column of 16 rows — column sums.
// TPW: 32-column tiles per wave (2; 1 for N <= 128, so that four waves still share the work).
template <int BLAY, bool SWAP, bool PRE, int TPW>
__global__ __launch_bounds__(64 * kRowsMaxWaves) void gemm_rows_kernel(RowsArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wid = tid >> 6, waves = nthreads >> 6;
    constexpr int WN = 32 * TPW;                               // columns per wave
    const int NP = waves * WN;                                 // columns covered by the waves' tiles (>= N)
    const int l31 = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.x * RM;
    float* As = lds;                                           // [RM][RK + 1]
    float* Bs = lds + RM * (RK + 1) + 8;                       // BLAY 0: [RK][NP]; BLAY 1: [NP][RK + 1]
    float* consts = Bs + ((BLAY == 0) ? RK * NP : NP * (RK + 1));      // PRE: [4][K] μ, invσ, dβ, dγ
    int* flag = reinterpret_cast<int*>(consts + (PRE ? 4 * g.K : 0));
    float* red = reinterpret_cast<float*>(flag + 4);           // epilogue scratch: [waves][RM] or [2][NP]

    if (PRE) {
        for (int k = tid; k < g.K; k += nthreads) {
            consts[k] = g.mean[k]; consts[g.K + k] = g.inv_std[k];
            const float db = static_cast<float>(g.bn_sums[k]), dg = static_cast<float>(g.bn_sums[g.K + k]);      // cudnn_utils.cu:158-173
            consts[2 * g.K + k] = db; consts[3 * g.K + k] = dg;
            if (blockIdx.x == 0) { g.dbeta[k] = db; g.dgamma[k] = dg; g.grad_bias[k] = db; }      // ∂β is the bias gradient; ∂γ is dropped (:173)
        }
        __syncthreads();
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- global -> register staging of one K tile ----
    // A: RM x RK = 256 float4, one per thread (threads >= 256): row tid >> 3, k offset 4 (tid & 7)
    // B: BLAY 0: RK rows x N/4 float4; BLAY 1: N rows x RK/4 float4 — strided over the threads
    const int a_row = tid >> 3, a_kq = (tid & 7) << 2;
    const bool a_mine = tid < 256;
    const int nb_f4 = (BLAY == 0) ? RK * (g.N >> 2) : g.N * (RK >> 2);
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rx = ra, rb[kRowsMaxBF4];
    auto load_tile = [&](int k0) {
        if (a_mine) {
            const int row = m0 + a_row, k = k0 + a_kq;
            const bool ok = row < g.M && k < g.K;                          // K % 4 == 0: a float4 is in or out as a whole
            const size_t off = ok ? static_cast<size_t>(row) * g.lda + k : 0;
            ra = *reinterpret_cast<const float4*>(g.A + off);
            if (PRE) rx = *reinterpret_cast<const float4*>(g.pre + off);
            if (!ok) ra = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < kRowsMaxBF4; ++it) {
            const int f = tid + it * nthreads;
            rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < nb_f4) {
                if (BLAY == 0) {
                    const int kk = f / (g.N >> 2), n4 = f - kk * (g.N >> 2);
                    if (k0 + kk < g.K) rb[it] = *reinterpret_cast<const float4*>(g.B + static_cast<size_t>(k0 + kk) * g.ldb + 4 * n4);
                } else {
                    const int n = f >> 3, kq = (f & 7) << 2;
                    if (k0 + kq < g.K) rb[it] = *reinterpret_cast<const float4*>(g.B + static_cast<size_t>(n) * g.ldb + k0 + kq);
                }
            }
        }
    };
    auto store_tile = [&](int k0) {
        if (a_mine) {
            if (PRE) {
                // dx = invσ · (dy − (dβ + x̂·dγ) / N), x̂ = (x − μ)·invσ      (bn_dx_kernel, loss_bn.hip)
                const int k = k0 + a_kq;
                if (k < g.K && m0 + a_row < g.M) {
                    float v[4] = {ra.x, ra.y, ra.z, ra.w}; const float x[4] = {rx.x, rx.y, rx.z, rx.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float is = consts[g.K + k + i];
                        const float xhat = (x[i] - consts[k + i]) * is;
                        v[i] = is * (v[i] - (consts[2 * g.K + k + i] + xhat * consts[3 * g.K + k + i]) * g.inv_n);
                    }
                    ra = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(g.A_rw + static_cast<size_t>(m0 + a_row) * g.lda + k) = ra;
                }
            }
            float* ap = As + a_row * (RK + 1) + a_kq;
            ap[0] = ra.x; ap[1] = ra.y; ap[2] = ra.z; ap[3] = ra.w;
        }
#pragma unroll
        for (int it = 0; it < kRowsMaxBF4; ++it) {
            const int f = tid + it * nthreads;
            if (f < nb_f4) {
                if (BLAY == 0) {
                    const int kk = f / (g.N >> 2), n4 = f - kk * (g.N >> 2);
                    *reinterpret_cast<float4*>(Bs + kk * NP + 4 * n4) = rb[it];
                } else {
                    const int n = f >> 3, kq = (f & 7) << 2;
                    float* bp = Bs + n * (RK + 1) + kq;
                    bp[0] = rb[it].x; bp[1] = rb[it].y; bp[2] = rb[it].z; bp[3] = rb[it].w;
                }
            }
        }
    };
    // columns N .. NP of the B image are never written by store_tile: zero them once (they feed MFMAs whose results are dropped,
    // but NaN bit patterns left in LDS would poison the row sums of squares otherwise)
    for (int i = tid; i < ((BLAY == 0) ? RK * NP : NP * (RK + 1)); i += nthreads) Bs[i] = 0.f;
    __syncthreads();

    load_tile(0);
    for (int k0 = 0; k0 < g.K; k0 += RK) {
        store_tile(k0);
        __syncthreads();
        if (k0 + RK < g.K) load_tile(k0 + RK);
#pragma unroll
        for (int kk = 0; kk < RK; kk += 2) {
            const int k = kk + lk;
            const float a = As[l31 * (RK + 1) + k];
            float b[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int n = wid * WN + j * 32 + l31;
                b[j] = (BLAY == 0) ? Bs[k * NP + n] : Bs[n * (RK + 1) + k];
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j)
                acc[j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a, acc[j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }

    if constexpr (SWAP) {
        // acc[j][4 q + t] = C[m0 + l31][wid WN + j 32 + 8 q + 4 lk + t]
        const int row = m0 + l31;
        float rsq = 0.f;
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wid * WN + j * 32 + 8 * q + 4 * lk;
                if (row < g.M && col < g.N) {                           // N % 4 == 0: the lane's four columns are in or out together
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = g.alpha * acc[j][4 * q + t] + (g.bias_n ? g.bias_n[col + t] : 0.f);
                    *reinterpret_cast<float4*>(g.C + static_cast<size_t>(row) * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) rsq += v[t] * v[t];
                }
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
    } else {
        // acc[j][r]: col = wid WN + j 32 + l31, row = m0 + (r & 3) + 8 (r >> 2) + 4 lk
        float cs[TPW], cs2[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            cs[j] = 0.f; cs2[j] = 0.f;
            const int col = wid * WN + j * 32 + l31;
            const float bias = (g.bias_n && col < g.N) ? g.bias_n[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < g.M && col < g.N) {
                    const float v = g.alpha * acc[j][r] + bias;
                    g.C[static_cast<size_t>(row) * g.ldc + col] = v;
                    cs[j] += v; cs2[j] += v * v;
                }
            }
        }
        if (g.colstats) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                cs[j] += __shfl_xor(cs[j], 32); cs2[j] += __shfl_xor(cs2[j], 32);
                if (lk == 0) { red[wid * WN + j * 32 + l31] = cs[j]; red[NP + wid * WN + j * 32 + l31] = cs2[j]; }
            }
            __syncthreads();
            const int N = g.N;
            double* out = g.colstats;
            grid_sum_ordered<0>(g.sums.part, g.sums.part2, g.sums.arrive, g.sums.fan, 2 * N, static_cast<int>(blockIdx.x),
                                static_cast<int>(gridDim.x), [&](int i) -> float { return i < N ? red[i] : red[NP + (i - N)]; },
                                [&](int i, double v) { out[i] = v; }, flag);
        }
    }
}

// LDS bytes of a launch, 0 when the shape is not covered: N <= 320, N and K multiples of 4, everything in 64 KB of LDS
static size_t rows_plan(int b_layout, int M, int N, int K, bool colstats, bool rowsq, bool bn, int* tpw_out, int* waves_out) {
    if (M <= 0 || N <= 0 || K <= 0 || N > 64 * kRowsMaxWaves || (N % 4) || (K % 4)) return 0;
    if ((colstats && rowsq) || (b_layout == 0 && (bn || rowsq)) || (bn && (b_layout != 1 || colstats))) return 0;
    const int tpw = N <= 128 ? 1 : 2;
    const int wn = 32 * tpw;
    const int waves = (N + wn - 1) / wn < 4 ? 4 : (N + wn - 1) / wn;  // >= 256 threads: the A tile is one float4 per thread
    const int threads = waves * 64, NP = waves * wn;
    const int nb_f4 = (b_layout == 0) ? RK * (N / 4) : N * (RK / 4);
    if ((nb_f4 + threads - 1) / threads > kRowsMaxBF4) return 0;
    const size_t b_img = (b_layout == 0) ? static_cast<size_t>(RK) * NP : static_cast<size_t>(NP) * (RK + 1);
    const size_t scratch = static_cast<size_t>(colstats ? 2 * NP : waves * RM);
    const size_t lds = (RM * (RK + 1) + 8 + b_img + (bn ? 4 * static_cast<size_t>(K) : 0) + 4 + scratch) * sizeof(float);
    if (lds > 64 * 1024) return 0;
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
    const size_t lds = rows_plan(b_layout, M, N, K, colstats != nullptr, rowsq != nullptr, bn != nullptr, &tpw, &waves);
    if (!lds) return false;
    const int threads = waves * 64;
    const int grid = (M + RM - 1) / RM;
    RowsArgs g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.bias_n = bias_n;
    g.colstats = colstats; g.rowsq = rowsq; g.rowsq_scale = rowsq_scale;
    if (colstats) {
        const int fan = grid_sum_fan(grid);
        if (!sums || sums->contrib_cap < grid || sums->width_cap < 2 * N || sums->groups_cap < (grid + fan - 1) / fan) return false;
        g.sums = *sums; g.sums.fan = fan;
    }
    if (bn) {
        g.A_rw = bn->dy; g.pre = bn->pre; g.mean = bn->mean; g.inv_std = bn->inv_std; g.bn_sums = bn->sums;
        g.dbeta = bn->dbeta; g.dgamma = bn->dgamma; g.grad_bias = bn->grad_bias; g.inv_n = static_cast<float>(1.0 / bn->n_global);
    }
    const dim3 grd(grid), blk(threads);
#define NVSM_ROWS_CASE(T)                                                                               \
    if (tpw == T) {                                                                                     \
        if (b_layout == 0) NVSM_LAUNCH((gemm_rows_kernel<0, false, false, T>), grd, blk, lds, s, g);    \
        else if (bn) NVSM_LAUNCH((gemm_rows_kernel<1, true, true, T>), grd, blk, lds, s, g);            \
        else NVSM_LAUNCH((gemm_rows_kernel<1, true, false, T>), grd, blk, lds, s, g);                   \
    }
    NVSM_ROWS_CASE(1) NVSM_ROWS_CASE(2)
#undef NVSM_ROWS_CASE
    return true;
}

int gemm_rows_max_m() {
    const char* e = std::getenv("NVSM_GEMM_ROWS_MAX");      // (read per call: tests and A/B runs switch it)
    return e ? std::atoi(e) : 16384;
}

}  // namespace cunvsm
