// gfx950 kernels: embedding gather-mean, fp32 MFMA GEMM, split-K reduce, small utilities.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "../../include/cunvsm_amd.h"
#include "kernels.h"
#include "device_utils.h"

namespace cunvsm {

// =============================================================================================
// gather-mean — replaces average_repr_kernel (cpp/params.cu:75-95), which runs one thread per
// output element with `window` dependent loads. Here the (row, 16-byte chunk) space is flattened
// so every lane issues independent, fully coalesced 16 B loads of consecutive chunks of one
// embedding row (a 300-float row = 75 chunks; a wave covers parts of at most two rows).
// out[b][t] = (Σ_j wt[b,j]·table[idx[b,j]][t]) / window     — divides by window even when weighted.
// =============================================================================================
// LAZY: the table decays lazily (kernels.h LazyView): a gathered row first gets the factors of the updates it sat out.
// U rows of the window are in flight per lane: their ids first, then the rows (and stamps) — a plain loop over the window
// compiles to id load → wait → row load → wait per word (twenty dependent round trips for a window of ten: 64 us where
// the bytes would take 45). The sum still runs over the window in order, one fused multiply-add per word.
struct NoLazyView { const int* stamp; int now; float decay[1]; };      // the eager kernel carries no 0.5 KB of factors
template <int V, bool LAZY, int U>
__global__ __launch_bounds__(256) void gather_mean_kernel(const float* __restrict__ table, int dim,
                                                          const int* __restrict__ idx,
                                                          const float* __restrict__ wts, int window,
                                                          uint32_t total, uint32_t nvec, float* __restrict__ out,
                                                          typename std::conditional<LAZY, LazyView, NoLazyView>::type lazy) {
#pragma clang fp contract(on)       // fused by the language rule, so that the LAZY and eager forms round alike
    const float fw = static_cast<float>(window);
    __shared__ float hist[LAZY ? kLazyHistory : 1];      // the factor history, out of the kernel arguments (per-lane index)
    if (LAZY) {
        for (int i = threadIdx.x; i < kLazyHistory; i += blockDim.x) hist[i] = lazy.decay[i];
        __syncthreads();
    }
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const uint32_t b = q / nvec;
        const uint32_t c = (q - b * nvec) * V;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        const int* ip = idx + static_cast<size_t>(b) * window;
        const float* wp = wts ? wts + static_cast<size_t>(b) * window : nullptr;
        for (int j0 = 0; j0 < window; j0 += U) {
            size_t row[U]; float wt[U], x[U][V]; int st[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(j0 + u, window - 1);           // (past the window: a harmless re-read, not added)
                row[u] = static_cast<size_t>(ip[j]);
                wt[u] = wp ? wp[j] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ldv<V>(table + row[u] * dim + c, x[u]);
                st[u] = LAZY ? lazy.stamp[row[u]] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j0 + u < window) {
                    if (LAZY) {
                        for (int k = st[u]; k < lazy.now; ++k) {
                            const float d = hist[k % kLazyHistory];
#pragma unroll
                            for (int i = 0; i < V; ++i) x[u][i] *= d;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[i] += wt[u] * x[u][i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = acc[i] / fw;
        stv<V>(out + static_cast<size_t>(b) * dim + c, acc);
    }
}

// rows in flight per lane: the whole window up to ten words, eight at a time beyond
int window_unroll(int window) { return window <= 10 ? (window < 1 ? 1 : window) : 8; }

template <int V, bool LAZY, typename View>
static void gather_mean_dispatch(int U, dim3 grid, hipStream_t s, const float* table, int dim, const int* idx, const float* wts,
                                 int window, uint32_t total, uint32_t nvec, float* out, const View& view) {
#define NVSM_GATHER_CASE(N) case N: NVSM_LAUNCH((gather_mean_kernel<V, LAZY, N>), grid, dim3(256), 0, s, table, dim, idx, wts, \
                                                       window, total, nvec, out, view); break;
    switch (U) {
        NVSM_GATHER_CASE(1) NVSM_GATHER_CASE(2) NVSM_GATHER_CASE(3) NVSM_GATHER_CASE(4) NVSM_GATHER_CASE(5)
        NVSM_GATHER_CASE(6) NVSM_GATHER_CASE(7) NVSM_GATHER_CASE(8) NVSM_GATHER_CASE(9) NVSM_GATHER_CASE(10)
        default: break;
    }
#undef NVSM_GATHER_CASE
}

void launch_gather_mean(const float* table, int dim, const int* idx, const float* wts, int window,
                        int64_t num_out, float* out, hipStream_t s, const LazyView* lazy) {
    if (num_out <= 0) return;
    const bool lz = lazy && lazy->stamp;
    // rows in flight per lane: the window in equal rounds of at most five (a window of ten: two rounds of five). All ten at once
    // was the first form of this loop — 70 / 90 registers, seven / five waves per SIMD —: five in flight and more waves are worth
    // 0.8 % of the 51 200-window step, 0.6 % at 6 400, 0.5 % at |D| = 2 M (2 the same; 3 / 4 / 6, whose last round is short: nothing)
    const int rounds = (window + 4) / 5;
    const int U = window < 1 ? 1 : (window + rounds - 1) / rounds;
    const bool vec = dim % 4 == 0;
    const uint32_t nvec = vec ? dim / 4 : dim;
    const uint32_t total = static_cast<uint32_t>(num_out * nvec);
    const dim3 grid(stream_grid(total, 256));
    if (vec) {
        if (lz) gather_mean_dispatch<4, true>(U, grid, s, table, dim, idx, wts, window, total, nvec, out, *lazy);
        else gather_mean_dispatch<4, false>(U, grid, s, table, dim, idx, wts, window, total, nvec, out, NoLazyView{});
    } else {
        if (lz) gather_mean_dispatch<1, true>(U, grid, s, table, dim, idx, wts, window, total, nvec, out, *lazy);
        else gather_mean_dispatch<1, false>(U, grid, s, table, dim, idx, wts, window, total, nvec, out, NoLazyView{});
    }
}

// =============================================================================================
// fp32 GEMM on the matrix cores: v_mfma_f32_32x32x2_f32 (exact f32, 157 TF/s peak on gfx950 — there
// is no TF32/xf32 path on CDNA4). Replaces the three cuBLAS sgemm calls of the path.
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 (2 x 2 MFMA tiles,
// 64 accumulator VGPRs), K step 32 staged through LDS. Operand fragments: lane l feeds
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the LDS images are laid out so that both
// fragment reads are bank-conflict-free ds_read_b32 (k-major image, or m-major image padded to 33).
// =============================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    int k_split_len;
    size_t c_split_stride;
    float alpha;
    const float* bias_n;
    double* colstats;        // optional [2][N]: Σ_rows C, Σ_rows C² of the stored values (batch-norm statistics, F6)
    GridSumWs sums;          // workspace of their ordered sum over the m tiles (one column group per n tile)
    float* rowsq;            // SWAP kernels, optional [ntiles][M]: rowsq_scale · Σ_{cols of the n tile} C² per row
    float rowsq_scale;
    int mtiles, ntiles, slabs, groups, members;   // see gemm_decode_block
};

// Global → register staging of one K tile (8 float4 per thread: 4 of A, 4 of B), zero-filled outside the
// matrix. FAST (every float4 is wholly inside or wholly outside: base 16 B aligned, leading dimensions and
// the contiguous extents multiples of 4): branch-free — the address is clamped to the matrix origin and the
// value selected to zero, so the eight loads issue back to back and stay in flight across the MFMA loop.
// !FAST: element-wise guarded loads (odd sizes in the tests).
template <int ALAY, int BLAY, bool FAST>
__device__ __forceinline__ void gemm_load_tile(const GemmArgs& g, int tid, int m0, int n0, int k0, int kend,
                                               float (&ra)[4][4], float (&rb)[4][4], unsigned& okmask) {
    okmask = 0;   // FAST: bit it (A) / bit 4+it (B) = this float4 lies inside the matrix; zeroing is deferred to
                  // gemm_store_tile so that nothing consumes the loads before the MFMA loop has run
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = tid + 256 * it;
        // ---- A ----
        {
            int r_, c_, rlim, clim;     // r_: index along the strided axis, c_: along the contiguous axis
            if constexpr (ALAY == 0) { r_ = m0 + (f >> 3); c_ = k0 + ((f & 7) << 2); rlim = g.M; clim = kend; }
            else                     { r_ = k0 + (f >> 5); c_ = m0 + ((f & 31) << 2); rlim = kend; clim = g.M; }
            if constexpr (FAST) {
                const bool ok = (r_ < rlim) && (c_ < clim);
                const size_t off = ok ? static_cast<size_t>(r_) * g.lda + c_ : 0;
                ldv<4>(g.A + off, ra[it]);
                okmask |= (ok ? 1u : 0u) << it;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[it][j] = 0.f;
                if (r_ < rlim) {
                    const float* p = g.A + static_cast<size_t>(r_) * g.lda + c_;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (c_ + j < clim) ra[it][j] = p[j];
                }
            }
        }
        // ---- B ----
        {
            int r_, c_, rlim, clim;
            if constexpr (BLAY == 0) { r_ = k0 + (f >> 5); c_ = n0 + ((f & 31) << 2); rlim = kend; clim = g.N; }
            else                     { r_ = n0 + (f >> 3); c_ = k0 + ((f & 7) << 2); rlim = g.N; clim = kend; }
            if constexpr (FAST) {
                const bool ok = (r_ < rlim) && (c_ < clim);
                const size_t off = ok ? static_cast<size_t>(r_) * g.ldb + c_ : 0;
                ldv<4>(g.B + off, rb[it]);
                okmask |= (ok ? 1u : 0u) << (4 + it);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) rb[it][j] = 0.f;
                if (r_ < rlim) {
                    const float* p = g.B + static_cast<size_t>(r_) * g.ldb + c_;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (c_ + j < clim) rb[it][j] = p[j];
                }
            }
        }
    }
}

template <int ALAY, int BLAY, bool FAST>
__device__ __forceinline__ void gemm_store_tile(int tid, float* As, float* Bs, float (&ra)[4][4], float (&rb)[4][4], unsigned okmask) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = tid + 256 * it;
        if constexpr (FAST) {
            const bool oka = (okmask >> it) & 1u, okb = (okmask >> (4 + it)) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[it][j] = oka ? ra[it][j] : 0.f; rb[it][j] = okb ? rb[it][j] : 0.f; }
        }
        if constexpr (ALAY == 0) {
            const int row = f >> 3, kq = (f & 7) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) As[row * (BK + 1) + kq + j] = ra[it][j];
        } else {
            const int kk = f >> 5, mq = (f & 31) << 2;
            stv<4>(As + kk * BM + mq, ra[it]);
        }
        if constexpr (BLAY == 0) {
            const int kk = f >> 5, nq = (f & 31) << 2;
            stv<4>(Bs + kk * BN + nq, rb[it]);
        } else {
            const int row = f >> 3, kq = (f & 7) << 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[row * (BK + 1) + kq + j] = rb[it][j];
        }
    }
}

// Block → (m tile, n tile, k slab). Tiles that share an operand slab (the n tiles of one m tile; all tiles of one
// split-K slab) get linear ids that are equal mod 8 and close together, i.e. they run at about the same time on
// the SAME XCD (observed placement: block b → XCD b % 8), so the shared slab is fetched from HBM once and served
// from that XCD's L2 afterwards. A different placement only costs speed.
__device__ __forceinline__ bool gemm_decode_block(const GemmArgs& g, int& mt, int& nt, int& z) {
    const int id = blockIdx.x;
    const int grp = id / (8 * g.members), rem = id - grp * (8 * g.members);
    const int member = rem >> 3, lane8 = rem & 7;
    const int group = grp * 8 + lane8;
    if (group >= g.groups) return false;
    if (g.slabs > 1) {              // group = k slab, member = (m tile, n tile)
        z = group; mt = member % g.mtiles; nt = member / g.mtiles;
    } else {                        // group = m tile, member = n tile
        z = 0; mt = group; nt = member;
    }
    return true;
}

// SWAP: the MFMA operands are fed swapped, i.e. each 32 x 32 tile is computed transposed, so that a lane ends up with
// 4 x 4 consecutive columns of ONE output row: 16 B stores (the epilogue is store-issue bound) and row-wise sums of
// squares for free (rowsq) — used for the dx·T GEMM, whose per-row mean of squares the Adam / Adagrad words update needs.
template <int ALAY, int BLAY, bool FAST, bool SWAP = false>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmArgs g) {
    constexpr int A_ELEMS = (ALAY == 0) ? BM * (BK + 1) : BK * BM;
    constexpr int B_ELEMS = (BLAY == 0) ? BK * BN : BN * (BK + 1);
    __shared__ __attribute__((aligned(16))) float lds[A_ELEMS + B_ELEMS + 8];
    float* As = lds;
    float* Bs = lds + ((A_ELEMS + 3) & ~3);

    int mt, nt, z;
    if (!gemm_decode_block(g, mt, nt, z)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, lk = lane >> 5;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = z * g.k_split_len;
    const int kend = min(g.K, kbeg + g.k_split_len);
    float* __restrict__ C = g.C + z * g.c_split_stride;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-column bias of this lane's two accumulator columns, fetched before the main loop
    float bias[2] = {0.f, 0.f};
    if (g.bias_n) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + l31;
            bias[j] = g.bias_n[col < g.N ? col : 0];
        }
    }

    float ra[4][4], rb[4][4];
    unsigned okmask;
    gemm_load_tile<ALAY, BLAY, FAST>(g, tid, m0, n0, kbeg, kend, ra, rb, okmask);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        gemm_store_tile<ALAY, BLAY, FAST>(tid, As, Bs, ra, rb, okmask);
        __syncthreads();
        // next tile's global loads fly while this tile is multiplied
        if (k0 + BK < kend) gemm_load_tile<ALAY, BLAY, FAST>(g, tid, m0, n0, k0 + BK, kend, ra, rb, okmask);

        // ---- 16 k-steps of 2; 4 MFMAs per step per wave ----
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int k = kk + lk;
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = wr * 64 + i * 32 + l31;
                a[i] = (ALAY == 0) ? As[m * (BK + 1) + k] : As[k * BM + m];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = wc * 64 + j * 32 + l31;
                b[j] = (BLAY == 0) ? Bs[k * BN + n] : Bs[n * (BK + 1) + k];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    if constexpr (SWAP) {
        // acc[i][j][4 q + t] = C[m0 + wr 64 + i 32 + l31][n0 + wc 64 + j 32 + 8 q + 4 lk + t]
        const bool vec_ok = (g.ldc % 4 == 0) && (reinterpret_cast<uintptr_t>(C) % 16 == 0);
        float rsq[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + wr * 64 + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = n0 + wc * 64 + j * 32 + 8 * q + 4 * lk;
                    if (row >= g.M || col >= g.N) continue;
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = g.alpha * acc[i][j][4 * q + t] + ((g.bias_n && col + t < g.N) ? g.bias_n[col + t] : 0.f);
                    float* cp = C + static_cast<size_t>(row) * g.ldc + col;
                    if (vec_ok && col + 3 < g.N) {
                        stv<4>(cp, v);
#pragma unroll
                        for (int t = 0; t < 4; ++t) rsq[i] += v[t] * v[t];
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (col + t < g.N) { cp[t] = v[t]; rsq[i] += v[t] * v[t]; }
                    }
                }
        }
        if (g.rowsq) {
            float* red = lds;             // [2 (wc)][BM]; the operand tiles are dead (loop ended on a barrier)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rsq[i] += __shfl_xor(rsq[i], 32);
                if (lk == 0) red[wc * BM + wr * 64 + i * 32 + l31] = rsq[i];
            }
            __syncthreads();
            if (tid < BM && m0 + tid < g.M) g.rowsq[static_cast<size_t>(nt) * g.M + m0 + tid] = (red[tid] + red[BM + tid]) * g.rowsq_scale;
        }
        return;
    }
    // ---- epilogue: C/D map col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
    // Interior tiles store unguarded: a per-row bounds branch would put an s_waitcnt vmcnt(0) (stores count in
    // vmcnt on CDNA4) between every pair of the 64 stores of a lane.
    // With colstats the per-column Σ and Σ² of the stored values ride along: per-lane fp32 sums over the lane's 32
    // rows, the two 32-lane halves folded by a cross-lane read, the two row-waves folded through LDS, then one fp64
    // atomic per column per block (replaces the separate column-statistics pass over the GEMM output).
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    float cs[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};
    if (interior) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* cp = C + static_cast<size_t>(m0 + wr * 64 + i * 32 + 4 * lk) * g.ldc + (n0 + wc * 64 + j * 32 + l31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = g.alpha * acc[i][j][r] + bias[j];
                    cp[static_cast<size_t>((r & 3) + 8 * (r >> 2)) * g.ldc] = v;
                    cs[j] += v; cs2[j] += v * v;
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wc * 64 + j * 32 + l31;
                if (col >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < g.M) {
                        const float v = g.alpha * acc[i][j][r] + bias[j];
                        C[static_cast<size_t>(row) * g.ldc + col] = v;
                        cs[j] += v; cs2[j] += v * v;
                    }
                }
            }
    }
    if (g.colstats) {
        float* red = lds;                 // [2 (Σ, Σ²)][2 (wr)][BN]; the operand tiles are dead (loop ended on a barrier)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cs[j] += __shfl_xor(cs[j], 32);
            cs2[j] += __shfl_xor(cs2[j], 32);
            if (lk == 0) {
                red[(0 * 2 + wr) * BN + wc * 64 + j * 32 + l31] = cs[j];
                red[(1 * 2 + wr) * BN + wc * 64 + j * 32 + l31] = cs2[j];
            }
        }
        __syncthreads();
        // ordered sum over the m tiles of this n tile (device_utils.h grid_sum_ordered): the same bits every run
        const GridSumWs& ws = g.sums;
        auto val = [&](int i) -> float { return i < BN ? red[i] + red[BN + i] : red[2 * BN + (i - BN)] + red[3 * BN + (i - BN)]; };
        double* cs = g.colstats;
        const int N = g.N;
        auto out = [&](int i, double v) {
            const int st = i / BN, n = i - st * BN;
            if (n0 + n < N) cs[static_cast<size_t>(st) * N + n0 + n] = v;
        };
        grid_sum_ordered<256>(ws.part + static_cast<size_t>(nt) * ws.contrib_cap * ws.width_cap,
                              ws.part2 + static_cast<size_t>(nt) * ws.groups_cap * ws.width_cap,
                              ws.arrive + nt * (ws.groups_cap + 1), ws.fan, 2 * BN, mt, g.mtiles, val, out,
                              reinterpret_cast<int*>(red + 4 * BN));
    }
}

bool launch_gemm_panel(int a_layout, int b_layout, const float* A, const float* B, float* C, int M, int N, int K,
                       int lda, int ldb, int ldc, float alpha, const float* bias_n, int slabs, int k_split_len,
                       size_t c_split_stride, hipStream_t s);   // gemm_panel.hip
static bool g_gemm_panel_forced_off = false;      // (process-wide test hook)
void gemm_set_panel_enabled(bool on) { g_gemm_panel_forced_off = !on; }

static int tiled_rowsq_parts(int N) { return (N + BN - 1) / BN; }
int gemm_rowsq_parts(int N) { return (N + 15) / 16; }

__global__ void sum_parts_kernel(const float* __restrict__ parts, int nparts, int64_t stride, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        // eight parts in flight, added in part order (the plain loop was a chain of nparts dependent round trips: 8 us
        // alone, 25-38 us on the critical stream next to the dT GEMM)
        float s = 0.f;
        for (int p0 = 0; p0 < nparts; p0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = parts[static_cast<int64_t>(min(p0 + u, nparts - 1)) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = (p0 + u == 0) ? v[u] : (p0 + u < nparts ? s + v[u] : s);
        }
        out[i] = s;
    }
}
void launch_sum_parts(const float* parts, int nparts, int64_t stride, float* out, int64_t n, hipStream_t s) {
    if (n > 0) NVSM_LAUNCH(sum_parts_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, parts, nparts, stride, out, n);
}

int gemm_split_k_slabs(int K, int want) {
    if (want <= 1) return 1;
    int len = (K + want - 1) / want;
    len = ((len + BK - 1) / BK) * BK;
    return (K + len - 1) / len;
}

void launch_gemm(int a_layout, int b_layout, const float* A, const float* B, float* C, int M, int N, int K,
                 int lda, int ldb, int ldc, float alpha, const float* bias_n, int split_k, size_t c_split_stride,
                 hipStream_t s, double* colstats, float* rowsq, float rowsq_scale, int* rowsq_parts, bool busy_chip,
                 const GridSumWs* sums, GemmSplitWs* split_ws) {
    if (M <= 0 || N <= 0) return;
    if (rowsq_parts) *rowsq_parts = rowsq ? tiled_rowsq_parts(N) : 0;
    // per-rank batch sizes: a workgroup per 32 rows and all columns (gemm_rows.hip); its row sums of squares are complete: one part
    if (split_k <= 1 && a_layout == 0 && M >= 512 && M <= gemm_rows_max_m() &&
        launch_gemm_rsplit(b_layout, A, B, C, M, N, K, lda, ldb, ldc, alpha, bias_n, s, colstats, sums, rowsq, rowsq_scale, split_ws, nullptr)) {
        if (rowsq_parts) *rowsq_parts = rowsq ? 1 : 0;
        return;
    }
    if (split_k <= 1 && a_layout == 0 && M >= 512 && M <= gemm_rows_max_m() &&
        launch_gemm_rows(b_layout, A, B, C, M, N, K, lda, ldb, ldc, alpha, bias_n, s, colstats, sums, rowsq, rowsq_scale, nullptr)) {
        if (rowsq_parts) *rowsq_parts = rowsq ? 1 : 0;
        return;
    }
    // large batches: the bf16 matrix pipe at fp32 accuracy (gemm_split.hip); complete row sums of squares: one part
    if (split_k <= 1 && a_layout == 0 && M > gemm_rows_max_m() &&
        launch_gemm_split(b_layout, A, B, C, M, N, K, lda, ldb, ldc, alpha, bias_n, s, colstats, sums, rowsq, rowsq_scale, split_ws)) {
        if (rowsq_parts) *rowsq_parts = rowsq ? 1 : 0;
        return;
    }
    // batch-sized products against the projection matrix: the matrix stationary in LDS (gemm_tstat.hip)
    if (split_k <= 1 && launch_gemm_tstat(a_layout, b_layout, A, B, C, M, N, K, lda, ldb, ldc, alpha, bias_n, s, colstats, rowsq,
                                          rowsq_scale, rowsq_parts, busy_chip, sums))
        return;
    GemmArgs g;
    g.colstats = (split_k > 1) ? nullptr : colstats;
    g.rowsq = nullptr; g.rowsq_scale = rowsq_scale;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.bias_n = bias_n; g.c_split_stride = c_split_stride;
    int slabs = 1;
    g.k_split_len = K;
    if (split_k > 1) {
        int len = (K + split_k - 1) / split_k;
        len = ((len + BK - 1) / BK) * BK;
        g.k_split_len = len;
        slabs = (K + len - 1) / len;
    }
    if (g.k_split_len <= 0) g.k_split_len = BK;
    // CU-sized panels for the large-batch projection shapes (see gemm_panel.hip); everything else: 128 x 128 tiles
    // (not next to the row passes of the fused step — `busy_chip`: a panel workgroup's four waves take a CU's whole register
    //  file, and a CU that hosts one hosts nothing else for its ~100 us; the tiled kernel takes 273 instead of 184 us there
    //  but shares its CUs: NVSM shape 1.007 -> 1.000 ms, full_adam 0.913 -> 0.902, |D| = 2 M 1.915 -> 1.90, interleaved A/B)
    if (!g_gemm_panel_forced_off && tuning().gemm_panel && !(busy_chip && split_k > 1) && launch_gemm_panel(a_layout, b_layout, A, B, C, M, N, K, lda, ldb, ldc, alpha, bias_n, slabs,
                                                  g.k_split_len, c_split_stride, s))
        return;
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(B) % 16 == 0);
    const int a_contig = a_layout == 0 ? K : M, b_contig = b_layout == 0 ? N : K;
    const bool fast = aligned && (a_contig % 4 == 0) && (b_contig % 4 == 0);
    g.mtiles = (M + BM - 1) / BM; g.ntiles = (N + BN - 1) / BN; g.slabs = slabs;
    if (g.colstats) {
        // (the caller sized the workspace for its largest batch; a launch it cannot hold is a programming error)
        const int fan = grid_sum_fan(g.mtiles);
        if (!sums || sums->colgroups < g.ntiles || sums->contrib_cap < g.mtiles || sums->width_cap < 2 * BN ||
            sums->groups_cap < (g.mtiles + fan - 1) / fan) {
            throw Error(NVSM_ERR_INVALID_ARGUMENT, "launch_gemm with column statistics needs a GridSumWs of " + std::to_string(g.ntiles) + " x " +
                                                           std::to_string(g.mtiles) + " x " + std::to_string(2 * BN));
        }
        g.sums = *sums;
        g.sums.fan = fan;
    }
    if (slabs > 1) { g.groups = slabs; g.members = g.mtiles * g.ntiles; }
    else { g.groups = g.mtiles; g.members = g.ntiles; }
    const int padded_groups = ((g.groups + 7) / 8) * 8;
    dim3 grid(padded_groups * g.members);
    dim3 block(256);
    // (Capping the residency at two workgroups per CU with unused dynamic LDS, to leave registers for the HBM-bound row
    // passes of the other stream, was A/B-tested interleaved: 1.268 vs 1.258 ms per step — no gain, not done.)
    constexpr size_t lds_pad = 0;
    if (rowsq && split_k <= 1) {
        if (fast && a_layout == 0 && b_layout == 1 && !g.colstats) {
            g.rowsq = rowsq;
            NVSM_LAUNCH((gemm_f32_mfma_kernel<0, 1, true, true>), grid, block, lds_pad, s, g);
            return;
        }
    }
#define NVSM_GEMM_CASE(AL, BL)                                                                              \
    if (a_layout == AL && b_layout == BL) {                                                                 \
        if (fast) NVSM_LAUNCH((gemm_f32_mfma_kernel<AL, BL, true>), grid, block, lds_pad, s, g);     \
        else NVSM_LAUNCH((gemm_f32_mfma_kernel<AL, BL, false>), grid, block, 0, s, g);               \
    }
    NVSM_GEMM_CASE(0, 0) NVSM_GEMM_CASE(0, 1) NVSM_GEMM_CASE(1, 0) NVSM_GEMM_CASE(1, 1)
#undef NVSM_GEMM_CASE
    if (rowsq && split_k <= 1) {      // shapes the SWAP kernel does not cover: a separate pass, all of it in part 0
        launch_row_meansq(C, M, N, rowsq_scale, rowsq, s);
        const int parts = tiled_rowsq_parts(N);
        if (parts > 1) (void)hipMemsetAsync(rowsq + M, 0, sizeof(float) * static_cast<size_t>(parts - 1) * M, s);
    }
}

// out[i] = Σ_z partial[z][i], z ascending within 16 interleaved groups that are then summed in group order (a fixed
// order: deterministic). 16 float4 columns x 16 slab groups per block so that ~1200 blocks stream the partials
// (the one-thread-per-output form ran 300 blocks with 128 dependent-free but serial loads each: 32 us for 39 MB).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int slabs, size_t stride,
                                                            float* __restrict__ out, int64_t n4) {
    __shared__ float4 red[16][16];
    const int cl = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int64_t c4 = static_cast<int64_t>(blockIdx.x) * 16 + cl;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < n4) {
        for (int z = sg; z < slabs; z += 16) {
            const float4 v = *reinterpret_cast<const float4*>(partial + z * stride + c4 * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[sg][cl] = acc;
    __syncthreads();
    if (sg == 0 && c4 < n4) {
        float4 t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][cl]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(out + c4 * 4) = t;
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_scalar_kernel(const float* __restrict__ partial, int slabs, size_t stride,
                                                                   float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < slabs; ++z) s += partial[z * stride + i];
        out[i] = s;
    }
}

void launch_splitk_reduce(const float* partial, int slabs, size_t stride, float* out, int64_t n, hipStream_t s) {
    const bool vec = (n % 4 == 0) && (stride % 4 == 0) && (reinterpret_cast<uintptr_t>(partial) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (vec) {
        const int64_t n4 = n / 4;
        NVSM_LAUNCH(splitk_reduce_kernel, dim3(static_cast<unsigned>((n4 + 15) / 16)), dim3(256), 0, s, partial, slabs, stride, out, n4);
    } else {
        NVSM_LAUNCH(splitk_reduce_scalar_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, partial, slabs, stride, out, n);
    }
}

// =============================================================================================
// small utilities
// =============================================================================================
// int64 ids of the ABI → int32, range-checked against the table they index (cpp/params.cu:75-95 and
// cpp/storage.cu:37-49 index the tables with them unchecked): an id outside [0, limit) is replaced by row 0 and
// `code` is stored into *err_flag — a word of page-locked host memory the engine inspects at its next
// synchronisation point (NVSM_ERR_INVALID_ARGUMENT). The store only happens on a bad id.
__global__ void narrow_i64_kernel(const int64_t* __restrict__ src, int* __restrict__ dst, int64_t n, int64_t limit,
                                  int* __restrict__ err_flag, int code) {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        int64_t v = src[i];
        if (static_cast<uint64_t>(v) >= static_cast<uint64_t>(limit)) {
            if (err_flag) *err_flag = code;
            v = 0;
        }
        dst[i] = static_cast<int>(v);
    }
}
void launch_narrow_i64(const int64_t* src, int* dst, int64_t n, int64_t limit, int* err_flag, int code, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(narrow_i64_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, src, dst, n, limit, err_flag, code);
}

// NVSM_DEBUG: the reference's CHECK_MATRIX (cpp/objective.cu:134,152 …) — any non-finite element stores `code`
__global__ void check_finite_kernel(const float* __restrict__ x, int64_t n, int* __restrict__ err_flag, int code) {
    bool bad = false;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        bad |= !isfinite(x[i]);
    if (bad) *err_flag = code;
}
void launch_check_finite(const float* x, int64_t n, int* err_flag, int code, hipStream_t s) {
    if (n > 0 && x) hipLaunchKernelGGL(check_finite_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, x, n, err_flag, code);
}

__global__ void scale_kernel(float* __restrict__ x, int64_t n, float a) {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        x[i] *= a;
}
void launch_scale(float* x, int64_t n, float a, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(scale_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, x, n, a);
}

// 16 B per lane and turn, four turns in flight: enough outstanding reads to fill the PCIe link from a few dozen workgroups
__global__ __launch_bounds__(256) void host_pull_kernel(HostPull p) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (int a = 0; a < p.count; ++a) {
        const size_t n16 = p.bytes[a] / 16;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ src = static_cast<const u32x4*>(p.src[a]);
        u32x4* __restrict__ dst = static_cast<u32x4*>(p.dst[a]);
        size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {
            const u32x4 v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
            const u32x4 v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
            dst[i] = v0; dst[i + stride] = v1; dst[i + 2 * stride] = v2; dst[i + 3 * stride] = v3;
        }
        for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
        // tail of fewer than 16 bytes (sizes are multiples of 4)
        const size_t done = n16 * 16, rest = (p.bytes[a] - done) / 4;
        if (blockIdx.x == 0 && threadIdx.x < rest)
            reinterpret_cast<uint32_t*>(static_cast<char*>(p.dst[a]) + done)[threadIdx.x] =
                reinterpret_cast<const uint32_t*>(static_cast<const char*>(p.src[a]) + done)[threadIdx.x];
    }
}
void launch_host_pull(const HostPull& p, hipStream_t s) {
    if (p.count <= 0) return;
    const int blocks = tuning().pull_blocks;      // (4-8 workgroups saturate the link; more only take wave slots from the step: 64 -> 8: 1.12 -> 1.075 ms)
    hipLaunchKernelGGL(host_pull_kernel, dim3(blocks), dim3(256), 0, s, p);
}

// Data parallel without synchronised batch-norm statistics: the step's ONE collective is the f32 all-reduce of
// [dT (d_e x d_w) | db (d_e) | loss hi | loss lo] — the bias gradient and the loss word ride behind the projection gradient instead
// of in an f64 all-reduce of their own (model.cpp backward_T). The loss word travels as two floats (hi = (float) s, lo = (float)
// (s - hi)): their sums, recombined in double, carry the fp64 word to ~1e-14 of its value per rank.
__global__ void dp_pack_tail_kernel(const float* __restrict__ gb, const double* __restrict__ loss, float* __restrict__ tail, int de) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < de) tail[i] = gb[i];
    if (i == de) {
        const double s = *loss;
        const float hi = static_cast<float>(s);
        tail[de] = hi;
        tail[de + 1] = static_cast<float>(s - static_cast<double>(hi));
    }
}
__global__ void dp_unpack_tail_kernel(const float* __restrict__ tail, float* __restrict__ gb, double* __restrict__ loss, int de) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < de) gb[i] = tail[i];
    if (i == de) *loss = static_cast<double>(tail[de]) + static_cast<double>(tail[de + 1]);
}
void launch_dp_pack_tail(const float* gb, const double* loss, float* tail, int de, hipStream_t s) {
    hipLaunchKernelGGL(dp_pack_tail_kernel, dim3((de + 1 + 255) / 256), dim3(256), 0, s, gb, loss, tail, de);
}
void launch_dp_unpack_tail(const float* tail, float* gb, double* loss, int de, hipStream_t s) {
    hipLaunchKernelGGL(dp_unpack_tail_kernel, dim3((de + 1 + 255) / 256), dim3(256), 0, s, tail, gb, loss, de);
}

__global__ void delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
void launch_delay(int microseconds, hipStream_t s) {
    if (microseconds > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, static_cast<long long>(microseconds) * 100);
}

__global__ void iota_kernel(int* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        dst[i] = static_cast<int>(i);
}
void launch_iota(int* dst, int64_t n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(iota_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, dst, n);
}

// SplitMix64 finaliser as a counter-based generator; ids uniform over [0, num_entities) via 64-bit
// multiply-high (bias < 2^-40). Same distribution as UniformLabelGenerator (cpp/labels.cu:4-22):
// slot 0 = the positive label, slots 1..k uniform over ALL documents (may repeat / hit the positive).
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// One launch for the start of a step in device-sampler mode (instead of memset + narrow + sample: four launches with
// their boundaries on the critical stream while nothing else is running): zero the step's statistics words, narrow
// the int64 word ids and draw the document ids (splitmix64 above, keyed by seed, step and b·R + r).
__global__ void step_prologue_kernel(const int64_t* __restrict__ words64, int* __restrict__ widx, int64_t nW,
                                     const int64_t* __restrict__ labels, int64_t N, int R, uint64_t num_words,
                                     uint64_t num_entities, uint64_t seed, uint64_t step, int* __restrict__ ids,
                                     double* __restrict__ stats, int nstats, int* __restrict__ err_flag, StampJob sj) {
    const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    if (tid < nstats) stats[tid] = 0.0;
    if (sj.list) {      // the previous words update's stamps (model.cpp lazy_end_update)
        const int64_t limit = *sj.count;
        for (int64_t i = tid; i < limit; i += stride) sj.stamp[sj.list[i]] = sj.value;
    }
    for (int64_t i = tid; i < nW; i += stride) {
        int64_t v = words64[i];
        if (static_cast<uint64_t>(v) >= num_words) { *err_flag = NVSM_BAD_WORD_ID; v = 0; }      // see narrow_i64_kernel
        widx[i] = static_cast<int>(v);
    }
    for (int64_t j = tid; j < N; j += stride) {
        const int64_t b = j / R;
        const int r = static_cast<int>(j - b * R);
        if (r == 0) {
            int64_t v = labels[b];
            if (static_cast<uint64_t>(v) >= num_entities) { *err_flag = NVSM_BAD_ENTITY_ID; v = 0; }
            ids[j] = static_cast<int>(v);
        } else {
            const uint64_t h = splitmix64(splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + static_cast<uint64_t>(j));
            ids[j] = static_cast<int>(__umul64hi(h, num_entities));
        }
    }
}
void launch_step_prologue(const int64_t* words64, int* widx, int64_t nW, const int64_t* labels, int64_t B, int R,
                          int64_t num_words, int64_t num_entities, uint64_t seed, uint64_t step, int* ids, double* stats,
                          int nstats, int* err_flag, hipStream_t s, StampJob stamps) {
    const int64_t N = B * R;
    int grid = stream_grid(N > nW ? N : nW, 256);
    const int need = (nstats + 255) / 256;
    if (grid < need) grid = need;
    NVSM_LAUNCH(step_prologue_kernel, dim3(grid), dim3(256), 0, s, words64, widx, nW, labels, N, R,
                       static_cast<uint64_t>(num_words), static_cast<uint64_t>(num_entities), seed, step, ids, stats, nstats,
                       err_flag, stamps);
}

}  // namespace cunvsm
