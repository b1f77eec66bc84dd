// fp32 MFMA GEMM, "panel" form, for the three projection GEMMs of the step at large batch: one workgroup per CU-sized
// output panel instead of 128 x 128 tiles. The tiled kernel (gather_gemm.hip) needs 800 workgroups for the forward
// projection at B = 51 200 while 768 fit on the chip at once (152 registers), so a 4 % tail runs alone and costs a
// third of the launch. Here a workgroup owns 16·TM rows x 64·TN columns (e.g. 208 x 256): 247 workgroups on 256 CUs,
// one each, every CU busy for the whole launch, 96 % of the MFMA slots doing useful work.
//
// v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue): A fragment lane l = A[i = l & 15][k = l >> 4], B fragment
// B[k = l >> 4][j = l & 15], C/D lane l holds rows 4 (l >> 4) + r, column l & 15 (the kernel feeds the operands
// swapped, i.e. computes transposed tiles, so that those 4 values are consecutive columns of one output row). Each of the 4 waves (one per SIMD,
// up to 512 registers) owns all TM row tiles x its own TN column tiles: TM·TN accumulators of 4 registers
// (13 x 4 → 208). Per k step a wave reads TM + TN operands from LDS for TM·TN MFMAs, so LDS bandwidth and bank
// conflicts are irrelevant; global → register → LDS staging of the next K tile overlaps the MFMAs of the current one.
#include "kernels.h"
#include "device_utils.h"

#include <cstdlib>

namespace cunvsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));


struct PanelArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    int k_split_len;           // K range per slab
    size_t c_split_stride;
    float alpha;
    const float* bias_n;
    int mpanels, npanels, slabs;
#ifdef NVSM_GEMM_DBG
    int dbg;                   // experiments only: 1 = one global tile load only, 2 = no C stores, 4 = no LDS tile stores
#endif
};

// (PBK = K tile depth)
// ALAY 0: A is [M][K] (k contiguous) → LDS As[m][PBK + 1];  ALAY 1: A stored [K][M] → LDS As[k][BM + 4]
// BLAY 0: B is [K][N] (n contiguous) → LDS Bs[k][BN + 4];   BLAY 1: B stored [N][K] → LDS Bs[n][PBK + 1]
template <int ALAY, int BLAY, int TM, int TN, int PBK>
__global__ __launch_bounds__(256, 1) void gemm_panel_kernel(PanelArgs g) {
    constexpr int KQ = PBK / 4;                                 // float4 per K tile row
    constexpr int BM = 16 * TM, BN = 64 * TN;
    constexpr int SA = (ALAY == 0) ? (PBK + 1) : (BM + 4);
    constexpr int SB = (BLAY == 0) ? (BN + 4) : (PBK + 1);
    constexpr int A_ELEMS = (ALAY == 0) ? BM * SA : PBK * SA;
    constexpr int B_ELEMS = (BLAY == 0) ? PBK * SB : BN * SB;
    constexpr int A_F4 = BM * PBK / 4, B_F4 = BN * PBK / 4;            // float4 per K tile
    constexpr int A_IT = (A_F4 + 255) / 256, B_IT = (B_F4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[((A_ELEMS + 3) & ~3) + B_ELEMS + 8];
    float* As = lds;
    float* Bs = lds + ((A_ELEMS + 3) & ~3);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int mp = bid % g.mpanels; bid /= g.mpanels;
    const int np = bid % g.npanels; const int z = bid / g.npanels;
    const int m0 = mp * BM, n0 = np * BN;
    const int kbeg = z * g.k_split_len, kend = min(g.K, kbeg + g.k_split_len);
    float* __restrict__ C = g.C + z * g.c_split_stride;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float ra[A_IT][4], rb[B_IT][4];
    // bit it (A) / bit 16 + it (B): this float4 lies inside the matrix. Out-of-range slots load a harmless in-range
    // address and are zeroed only in store_tile, so nothing consumes the loads before the MFMA loop has run.
    unsigned okmask = 0;
    auto load_tile = [&](int k0) {
        okmask = 0;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int f = tid + 256 * it;
            int r_, c_, rlim, clim;
            if (ALAY == 0) { r_ = m0 + f / KQ; c_ = k0 + ((f % KQ) << 2); rlim = g.M; clim = kend; }
            else           { r_ = k0 + f / (BM / 4); c_ = m0 + ((f % (BM / 4)) << 2); rlim = kend; clim = g.M; }
            const bool ok = (f < A_F4) && (r_ < rlim) && (c_ < clim);
            const size_t off = ok ? static_cast<size_t>(r_) * g.lda + c_ : 0;
            ldv<4>(g.A + off, ra[it]);
            okmask |= (ok ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int f = tid + 256 * it;
            int r_, c_, rlim, clim;
            if (BLAY == 0) { r_ = k0 + f / (BN / 4); c_ = n0 + ((f % (BN / 4)) << 2); rlim = kend; clim = g.N; }
            else           { r_ = n0 + f / KQ; c_ = k0 + ((f % KQ) << 2); rlim = g.N; clim = kend; }
            const bool ok = (f < B_F4) && (r_ < rlim) && (c_ < clim);
            const size_t off = ok ? static_cast<size_t>(r_) * g.ldb + c_ : 0;
            ldv<4>(g.B + off, rb[it]);
            okmask |= (ok ? 1u : 0u) << (16 + it);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int f = tid + 256 * it;
            if (f >= A_F4) continue;
            const bool oka = (okmask >> it) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[it][j] = oka ? ra[it][j] : 0.f;
            if (ALAY == 0) {
                const int row = f / KQ, kq = (f % KQ) << 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) As[row * SA + kq + j] = ra[it][j];
            } else {
                const int kk = f / (BM / 4), mq = (f % (BM / 4)) << 2;
                stv<4>(As + kk * SA + mq, ra[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int f = tid + 256 * it;
            if (f >= B_F4) continue;
            const bool okb = (okmask >> (16 + it)) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[it][j] = okb ? rb[it][j] : 0.f;
            if (BLAY == 0) {
                const int kk = f / (BN / 4), nq = (f % (BN / 4)) << 2;
                stv<4>(Bs + kk * SB + nq, rb[it]);
            } else {
                const int row = f / KQ, kq = (f % KQ) << 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[row * SB + kq + j] = rb[it][j];
            }
        }
    };

    load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += PBK) {
#ifdef NVSM_GEMM_DBG
        if (!(g.dbg & 4) || k0 == kbeg)
#endif
        store_tile();
        __syncthreads();
#ifdef NVSM_GEMM_DBG
        if (k0 + PBK < kend && !(g.dbg & 1)) load_tile(k0 + PBK);
#else
        if (k0 + PBK < kend) load_tile(k0 + PBK);      // next tile's global loads fly while this tile is multiplied
#endif
#pragma unroll
        for (int ks = 0; ks < PBK; ks += 4) {
            const int k = ks + lg;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = i * 16 + li;
                a[i] = (ALAY == 0) ? As[m * SA + k] : As[k * SA + m];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = (wid * TN + j) * 16 + li;
                b[j] = (BLAY == 0) ? Bs[k * SB + n] : Bs[n * SB + k];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // operands swapped: the MFMA computes the TRANSPOSED tile, so a lane ends up with 4 consecutive
                    // columns of one row — a 16 B store instead of four 4 B stores (the epilogue is store-issue bound)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: acc[i][j][r] = C[m0 + 16 i + li][n0 + 16 (wid TN + j) + 4 lg + r] ----
    const bool vec_ok = (g.ldc % 4 == 0) && (reinterpret_cast<uintptr_t>(C) % 16 == 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wid * TN + j) * 16 + 4 * lg;
        float bias[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { bias[r] = (g.bias_n && col + r < g.N) ? g.bias_n[col + r] : 0.f; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + i * 16 + li;
            if (row >= g.M || col >= g.N) continue;
#ifdef NVSM_GEMM_DBG
            if ((g.dbg & 2) && acc[i][j][0] != 12345.678f) continue;
#endif
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = g.alpha * acc[i][j][r] + bias[r];
            float* cp = C + static_cast<size_t>(row) * g.ldc + col;
            if (vec_ok && col + 3 < g.N) {
                stv<4>(cp, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < g.N) cp[r] = v[r];
            }
        }
    }
}

template <int ALAY, int BLAY, int TM, int TN, int PBK>
static void launch_panel(const PanelArgs& g, hipStream_t s) {
    const int grid = g.mpanels * g.npanels * g.slabs;
    NVSM_LAUNCH((gemm_panel_kernel<ALAY, BLAY, TM, TN, PBK>), dim3(grid), dim3(256), 0, s, g);
}

// Returns true when the panel kernel took the GEMM; false → the caller uses the tiled kernel.
bool launch_gemm_panel(int a_layout, int b_layout, const float* A, const float* B, float* C, int M, int N, int K,
                       int lda, int ldb, int ldc, float alpha, const float* bias_n, int slabs, int k_split_len,
                       size_t c_split_stride, hipStream_t s) {
    PanelArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.bias_n = bias_n;
    g.slabs = slabs; g.k_split_len = slabs > 1 ? k_split_len : K; g.c_split_stride = c_split_stride;
    const int a_contig = a_layout == 0 ? K : M, b_contig = b_layout == 0 ? N : K;
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(B) % 16 == 0) && (a_contig % 4 == 0) && (b_contig % 4 == 0);
    if (!aligned) return false;
    // (A forward-product variant — TM 13 or 7, one or two workgroups per CU: 88 us alone against 102 us tiled at B = 51 200 —
    //  lost inside the step: a grid that needs every workgroup slot of the chip gets a second scheduling round as soon as
    //  anything else is resident, 136 us. Removed in round 3 together with its atomic column statistics.)
    if (a_layout == 1 && b_layout == 0 && slabs >= 64 && N <= 256 && N > 128 && M <= 320 && M > 160) {  // dT, split-K slabs
        g.mpanels = 2; g.npanels = 1;
        launch_panel<1, 0, 10, 4, 32>(g, s);                    // (2 row halves) x slabs workgroups, 160 accumulators
        return true;
    }
    return false;
}

}  // namespace cunvsm
