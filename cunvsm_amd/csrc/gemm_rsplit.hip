// The two batch-sized projection products at PER-RANK batch sizes (512 ... 8 192 rows: the 6 400 windows a rank of the 8-GPU
// job holds, the LSE recipe's 4 096) on the bf16 matrix pipe at fp32 accuracy:
//   forward   pre[B][d_e]     = phrase[B][d_w] · Tt[d_w][d_e]  (+ bias, + batch-norm column sums)   cpp/params.cu:417
//   backward  gphrase[B][d_w] = alpha · dx[B][d_e] · T          (+ per-row mean of squares,          cpp/objective.cu:453
//                                                                + the batch-norm backward on the way in: cpp/cudnn_utils.cu:143-183)
// Round 3's row-panel kernel (gemm_rows.hip: exact-fp32 MFMAs, K in tiles of 32 through two LDS images) took 28 / 24 us alone for
// 0.98 GFLOP at 6 400 rows — 17-21 % of the fp32 MFMA peak — and none of it was the matrix pipe's fault: ten K tiles, each a
// global -> register -> LDS -> barrier -> fragment hop of 1.5-2 us that nothing but the same workgroup's own MFMAs could hide
// (200 workgroups on 256 CUs: one round, no second workgroup to switch to). This kernel keeps the decomposition (a workgroup owns
// 32 rows and ALL columns, so the batch-norm backward rides on the load and the row statistics come out complete) and removes
// the hops:
//   * the workgroup's WHOLE panel of A (32 x K fp32, 38 KB at K = 300) is requested at once, cut into its three bf16 planes
//     (gemm_split.hip: x = h + m + l exactly) and stored to LDS in MFMA fragment order — one round trip to memory and ONE barrier
//     per workgroup instead of one per K tile;
//   * B — the projection matrix, the same for every workgroup — arrives already cut and in fragment order (gemm_rsplit_planes_
//     kernel, behind the projection update, off the critical path); a wave owns one 32-column tile and fetches its three 1 KB
//     fragments per k step straight from L2, two k steps ahead;
//   * the K loop is barrier-free: per k step of 16 three ds_read_b128 (conflict-free: a k step's fragments are 64 lanes x 16 B
//     in lane order, XOR-swizzled by k step so that the staging stores do not collide either), three 16 B global loads and six
//     v_mfma_f32_32x32x16_bf16 (nine with NVSM_GEMM_SPLIT=9) into two accumulators in rotation;
//   * operands fed swapped (the tile is computed transposed): a lane owns 4 x 4 consecutive columns of one output row — 16 B
//     stores, row sums of squares by one cross-lane add; the epilogues are gemm_rows.hip's (ordered column sums in a fixed order).
// Six bf16 MFMAs per fp32 product run in 6 / 16 of the fp32 pipe's time; what the kernel costs is launch + one memory round trip +
// K / 16 x 6 MFMAs + epilogue. Accuracy: as gemm_split.hip (tests/test_gpu_parity.py::test_gemm_split_bf16_is_fp32_accurate
// covers both kernels).
#include "kernels.h"
#include "device_utils.h"

#include <atomic>
#include <cstdlib>

namespace cunvsm {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SM = 32;                          // rows per workgroup
constexpr size_t kRsLdsMax = 150 * 1024;
constexpr int kRsMaxDevices = 64;

struct RsArgs {
    const float* A; float* C;
    int M, N, K, lda, ldc;
    int KSP;                                   // k steps of 16, padded to an even number
    int NT;                                    // 32-column tiles
    float alpha;
    const float* bias_n;
    double* colstats; GridSumWs sums;          // forward: [2][N] = Σ_rows C, Σ_rows C² (ordered grid-wide sum)
    float* rowsq; float rowsq_scale;           // backward: rowsq[M] = rowsq_scale · Σ_cols C²
    const unsigned char* planes;               // B: [3][KSP][NT][64 lanes][16 B]
    // PRE: batch-norm backward on the rows of A (= dy, overwritten with dx)
    float* A_rw; const float* pre; const float* mean; const float* inv_std; const double* bn_sums;
    float* dbeta; float* dgamma; float* grad_bias; float inv_n;
    float* dump;
};

// GATH: the panel of A is not read but MADE — the word gather-mean of the forward pass (gather_gemm.hip gather_mean_kernel,
// cpp/params.cu:75-95) rides on the staging: a thread that would have fetched float4 c4 of phrase row r fetches that float4 of
// the window's word rows instead (the 32 windows' ids and weights first, into LDS; then five rows in flight per float4), adds them
// up in the gather kernel's order — acc = fma(w_j, x_j, acc) for j ascending, then acc / window — and writes the result to
// `phrase` on the way (the dT product reads it later). Same bits as the gather kernel's, one launch, one gap and a 2 x B x K x 4
// byte round trip through memory less on the step's critical stream. GATH 2: the words table decays lazily (kernels.h LazyView).
constexpr int kRsGatherU = 5;                  // rows in flight per float4 of the panel
constexpr int kRsGatherMaxWindow = 32;
struct RsNoGather { };
struct RsGather {
    const float* table; const int* idx; const float* wts; int window; float* phrase;
};
struct RsGatherLazy {
    const float* table; const int* idx; const float* wts; int window; float* phrase;
    LazyView lazy;
};
template <int GATH> struct RsGatherArgs { typedef RsNoGather type; };
template <> struct RsGatherArgs<1> { typedef RsGather type; };
template <> struct RsGatherArgs<2> { typedef RsGatherLazy type; };

// a - b as ONE v_sub_f32 (see gemm_split.hip split_sub: packed fp32 VALU is slow next to MFMAs)
__device__ __forceinline__ float rs_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x0, x1 -> one 32-bit word per plane (x0's piece in the lower half): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), all
// round-to-nearest; both differences are exact in fp32 and the last has at most eight significant bits: x = h + m + l exactly
__device__ __forceinline__ void rs_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
    const float r0 = rs_sub(x0, __uint_as_float(h << 16)), r1 = rs_sub(x1, __uint_as_float(h & 0xffff0000u));
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
    const float s0 = rs_sub(r0, __uint_as_float(m << 16)), s1 = rs_sub(r1, __uint_as_float(m & 0xffff0000u));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}

__device__ __forceinline__ f32x16 rs_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Where the fragment of (k step ks, lane) sits inside a plane: 1 KB per k step, the lane's 16 B at slot lane ^ swz(ks, half).
// A ds_read_b128 is serviced in four groups of sixteen lanes, all inside one half (lanes < 32 / >= 32), and an XOR of the low
// three lane bits permutes a group's sixteen slots among themselves mod 16: every group still covers all 64 banks once. The
// staging stores of one thread group, on the other hand, walk the k octets of ONE row — (ks, half) changes, the row does not —
// and without the swizzle all of them fall on the same four banks.
__device__ __forceinline__ int rs_slot(int ks, int lane) { return lane ^ (((ks & 3) << 1) | (lane >> 5)); }

struct RsFrag { u32x4 h, m, l; };

// RCH: float4s of the panel a thread holds at once while staging (the panel is 32 x 4 KSP float4s over 64 WAVES threads)
template <bool PRE, int WAVES, int RCH, int NPROD, int GATH = 0>
__global__ __launch_bounds__(64 * WAVES) void gemm_rsplit_kernel(RsArgs g, typename RsGatherArgs<GATH>::type ga) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    constexpr int T = 64 * WAVES;
    constexpr int NP = 32 * WAVES;                              // columns covered by the waves' tiles (>= N)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.x * SM;
    const int KSP = g.KSP;
    const int plane_bytes = KSP * 1024;
    // LDS: three planes of A (dead behind the K loop: the column-sum tile [32][NP + 4] takes their place) | scratch | constants
    constexpr int TP = NP + 4;
    const int img_bytes = (3 * plane_bytes > SM * TP * 4) ? 3 * plane_bytes : SM * TP * 4;
    float* red = reinterpret_cast<float*>(rs_lds + img_bytes);  // [WAVES][32] or [2][NP]
    float* consts = red + 2 * NP;                               // PRE: [4][K] μ, invσ, dβ, dγ
    int* flag = reinterpret_cast<int*>(consts + (PRE ? 4 * g.K : 0));
    // GATH: [32 x window] ids | weights | (lazy) stamps | (lazy) the factor history
    int* gid = flag + 4;

    // ---- this wave's fragments of B for the first two k steps: requested before anything else ----
    const bool has_tile = wid < g.NT;
    const size_t bplane = static_cast<size_t>(KSP) * g.NT * 1024;
    const unsigned char* bbase = g.planes + (static_cast<size_t>(has_tile ? wid : 0) * 64 + lane) * 16;
    auto load_b = [&](int ks, RsFrag& f) {
        const unsigned char* p = bbase + static_cast<size_t>(ks < KSP ? ks : KSP - 1) * g.NT * 1024;
        f.h = *reinterpret_cast<const u32x4*>(p);
        f.m = *reinterpret_cast<const u32x4*>(p + bplane);
        f.l = *reinterpret_cast<const u32x4*>(p + 2 * bplane);
    };
    RsFrag b0, b1;
    load_b(0, b0);
    load_b(1, b1);

    if (PRE) {
        for (int k = tid; k < g.K; k += T) {
            consts[k] = g.mean[k]; consts[g.K + k] = g.inv_std[k];
            const float db = static_cast<float>(g.bn_sums[k]), dg = static_cast<float>(g.bn_sums[g.K + k]);      // cudnn_utils.cu:158-173
            consts[2 * g.K + k] = db; consts[3 * g.K + k] = dg;
            if (blockIdx.x == 0) { g.dbeta[k] = db; g.dgamma[k] = dg; g.grad_bias[k] = db; }      // ∂β is the bias gradient; ∂γ is dropped (:173)
        }
        __syncthreads();
    }
    if (!PRE && g.grad_bias && blockIdx.x == 0)      // no batch-norm: the bias gradient is Σdy (launch_colsum_finalize's job, riding here)
        for (int k = tid; k < g.K; k += T) g.grad_bias[k] = static_cast<float>(g.bn_sums[k]);

    // ---- stage the panel: 32 rows x 4 KSP float4s (zeros past K and past M), RCH per thread at a time ----
    if constexpr (GATH != 0) {
        const int window = ga.window;
        const int nid = SM * window;
        float* gwt = reinterpret_cast<float*>(gid + nid);
        int* gst = reinterpret_cast<int*>(gwt + nid);
        float* hist = reinterpret_cast<float*>(gst + nid);
        // the 32 windows' ids and weights (rows past M: row 0 with weight 0 — loaded, never stored)
        for (int i = tid; i < nid; i += T) {
            const int r = i / window;
            const bool ok = m0 + r < g.M;
            const size_t at = static_cast<size_t>(m0) * window + i;
            gid[i] = ok ? ga.idx[at] : 0;
            gwt[i] = ok ? (ga.wts ? ga.wts[at] : 1.f) : 0.f;
        }
        if constexpr (GATH == 2)
            for (int i = tid; i < kLazyHistory; i += T) hist[i] = ga.lazy.decay[i];
        __syncthreads();
        const int C4 = 4 * KSP, K4 = g.K >> 2;
        const int total = SM * C4;
        const float fw = static_cast<float>(window);
        for (int base = 0; base < total; base += T * RCH) {
            int row[RCH], c4[RCH]; bool in[RCH];
            float acc[RCH][4];
#pragma unroll
            for (int u = 0; u < RCH; ++u) {
                const int idx = base + tid + T * u;
                row[u] = idx / C4; c4[u] = idx - row[u] * C4;
                in[u] = idx < total && c4[u] < K4 && m0 + row[u] < g.M;
                if (!(idx < total)) row[u] = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u][e] = 0.f;
            }
            for (int j0 = 0; j0 < window; j0 += kRsGatherU) {
                u32x4 x[RCH][kRsGatherU];
#pragma unroll
                for (int u = 0; u < RCH; ++u) {
#pragma unroll
                    for (int v = 0; v < kRsGatherU; ++v) {
                        const int j = min(j0 + v, window - 1);          // (past the window: a harmless re-read, not added)
                        const int id = gid[row[u] * window + j];
                        const size_t off = in[u] ? static_cast<size_t>(id) * g.lda + 4 * c4[u] : 0;
                        x[u][v] = *reinterpret_cast<const u32x4*>(ga.table + off);
                    }
                }
                if constexpr (GATH == 2) {
                    if (base == 0 && j0 == 0) {      // the stamps of the windows' rows: requested behind the first round of rows, needed in front of the first sum
                        for (int i = tid; i < nid; i += T) gst[i] = ga.lazy.stamp[gid[i]];
                        __syncthreads();
                    }
                }
#pragma unroll
                for (int u = 0; u < RCH; ++u) {
#pragma unroll
                    for (int v = 0; v < kRsGatherU; ++v) {
                        if (j0 + v < window) {
                            const int at = row[u] * window + j0 + v;
                            float xf[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) xf[e] = __uint_as_float(x[u][v][e]);
                            if constexpr (GATH == 2) {
                                for (int k = gst[at]; k < ga.lazy.now; ++k) {
                                    const float d = hist[k % kLazyHistory];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) xf[e] *= d;
                                }
                            }
                            const float wt = gwt[at];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[u][e] = __builtin_fmaf(wt, xf[e], acc[u][e]);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RCH; ++u) {
                const int idx = base + tid + T * u;
                if (idx >= total) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = in[u] ? acc[u][e] / fw : 0.f;
                if (in[u]) *reinterpret_cast<float4*>(ga.phrase + static_cast<size_t>(m0 + row[u]) * g.lda + 4 * c4[u]) = make_float4(v[0], v[1], v[2], v[3]);
                unsigned h0, m0_, l0, h1, m1, l1;
                rs_pair(v[0], v[1], h0, m0_, l0);
                rs_pair(v[2], v[3], h1, m1, l1);
                const int ks = c4[u] >> 2, fl = row[u] + 32 * ((c4[u] >> 1) & 1);
                unsigned char* p = rs_lds + ks * 1024 + rs_slot(ks, fl) * 16 + (c4[u] & 1) * 8;
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + plane_bytes) = make_uint2(m0_, m1);
                *reinterpret_cast<uint2*>(p + 2 * plane_bytes) = make_uint2(l0, l1);
            }
        }
    } else {
        const int C4 = 4 * KSP, K4 = g.K >> 2;
        const int total = SM * C4;
        for (int base = 0; base < total; base += T * RCH) {
            u32x4 av[RCH], xv[RCH];
            int row[RCH], c4[RCH]; bool in[RCH];
#pragma unroll
            for (int u = 0; u < RCH; ++u) {
                const int idx = base + tid + T * u;
                row[u] = idx / C4; c4[u] = idx - row[u] * C4;
                in[u] = idx < total && c4[u] < K4 && m0 + row[u] < g.M;
                const size_t off = in[u] ? static_cast<size_t>(m0 + row[u]) * g.lda + 4 * c4[u] : 0;
                av[u] = *reinterpret_cast<const u32x4*>(g.A + off);
                if (PRE) xv[u] = *reinterpret_cast<const u32x4*>(g.pre + off);
            }
#pragma unroll
            for (int u = 0; u < RCH; ++u) {
                const int idx = base + tid + T * u;
                if (idx >= total) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = in[u] ? __uint_as_float(av[u][e]) : 0.f;
                if (PRE) {
                    // dx = invσ · (dy − (dβ + x̂·dγ) / N), x̂ = (x − μ)·invσ      (bn_dx_kernel, loss_bn.hip); written back over dy
                    const int k = in[u] ? 4 * c4[u] : 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float is = consts[g.K + k + e];
                        const float xhat = (__uint_as_float(xv[u][e]) - consts[k + e]) * is;
                        v[e] = is * (v[e] - (consts[2 * g.K + k + e] + xhat * consts[3 * g.K + k + e]) * g.inv_n);
                    }
                    *reinterpret_cast<float4*>(in[u] ? g.A_rw + static_cast<size_t>(m0 + row[u]) * g.lda + 4 * c4[u] : g.dump) =
                        make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = in[u] ? v[e] : 0.f;
                }
                unsigned h0, m0_, l0, h1, m1, l1;
                rs_pair(v[0], v[1], h0, m0_, l0);
                rs_pair(v[2], v[3], h1, m1, l1);
                // float4 c4 of a row: k step c4 / 4, k octet (c4 / 2) & 1 = the lane half, first or second 8 bytes of the fragment
                const int ks = c4[u] >> 2, fl = row[u] + 32 * ((c4[u] >> 1) & 1);
                unsigned char* p = rs_lds + ks * 1024 + rs_slot(ks, fl) * 16 + (c4[u] & 1) * 8;
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + plane_bytes) = make_uint2(m0_, m1);
                *reinterpret_cast<uint2*>(p + 2 * plane_bytes) = make_uint2(l0, l1);
            }
        }
    }
    __syncthreads();

    // ---- K loop: no barriers. acc0 / acc1 in rotation (a chain of dependent MFMAs runs at the pipe's latency, not its rate) ----
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    auto read_a = [&](int ks, RsFrag& f) {
        const int kk = ks < KSP ? ks : KSP - 1;
        const unsigned char* p = rs_lds + kk * 1024 + rs_slot(kk, lane) * 16;
        f.h = *reinterpret_cast<const u32x4*>(p);
        f.m = *reinterpret_cast<const u32x4*>(p + plane_bytes);
        f.l = *reinterpret_cast<const u32x4*>(p + 2 * plane_bytes);
    };
    auto multiply = [&](const RsFrag& b, const RsFrag& a) {
        // smallest products first; the operands swapped: the tile comes out transposed (a lane owns columns, see the epilogue)
        if (NPROD == 9) {
            acc0 = rs_mfma(b.l, a.l, acc0);
            acc1 = rs_mfma(b.l, a.m, acc1);
            acc0 = rs_mfma(b.m, a.l, acc0);
        }
        acc1 = rs_mfma(b.l, a.h, acc1);
        acc0 = rs_mfma(b.h, a.l, acc0);
        acc1 = rs_mfma(b.m, a.m, acc1);
        acc0 = rs_mfma(b.m, a.h, acc0);
        acc1 = rs_mfma(b.h, a.m, acc1);
        acc0 = rs_mfma(b.h, a.h, acc0);
    };
    if (has_tile) {
        RsFrag a0, a1;
        read_a(0, a0);
        for (int ks = 0; ks < KSP; ks += 2) {          // (KSP is even; loads past the end repeat the last k step and are not used)
            read_a(ks + 1, a1);
            multiply(b0, a0);
            load_b(ks + 2, b0);
            read_a(ks + 2, a0);
            multiply(b1, a1);
            load_b(ks + 3, b1);
        }
    }
    __syncthreads();                                    // the planes are dead: the epilogue's tile may overwrite them

    // ---- epilogue (gemm_rows.hip's): acc[4 q + t] = C[m0 + l31][32 wid + 8 q + 4 lk + t] ----
    const int row = m0 + l31;
    float* tile = reinterpret_cast<float*>(rs_lds);
    float rsq = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = wid * 32 + 8 * q + 4 * lk;
        const bool ok = has_tile && row < g.M && col < g.N;             // N % 4 == 0: the lane's four columns are in or out together
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = ok ? g.alpha * (acc0[4 * q + t] + acc1[4 * q + t]) + (g.bias_n ? g.bias_n[col + t] : 0.f) : 0.f;
        if (ok) *reinterpret_cast<float4*>(g.C + static_cast<size_t>(row) * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) rsq += v[t] * v[t];
        if (g.colstats) *reinterpret_cast<float4*>(tile + l31 * TP + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (g.rowsq) {
        rsq += __shfl_xor(rsq, 32);
        if (lk == 0) red[wid * SM + l31] = rsq;
        __syncthreads();
        if (tid < SM && m0 + tid < g.M) {
            float s = 0.f;
            for (int w = 0; w < WAVES; ++w) s += red[w * SM + tid];       // wave order: the same sum every run
            g.rowsq[m0 + tid] = s * g.rowsq_scale;
        }
    }
    if (g.colstats) {
        __syncthreads();
        for (int c = tid; c < NP; c += T) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
            for (int r = 0; r < SM; ++r) { const float v = tile[r * TP + c]; s1 += v; s2 += v * v; }      // (rows >= M hold zeros)
            red[c] = s1; red[NP + c] = s2;
        }
        __syncthreads();
        const int N = g.N;
        double* out = g.colstats;
        grid_sum_ordered<T>(g.sums.part, g.sums.part2, g.sums.arrive, g.sums.fan, 2 * N, static_cast<int>(blockIdx.x),
                            static_cast<int>(gridDim.x), [&](int i) -> float { return i < N ? red[i] : red[NP + (i - N)]; },
                            [&](int i, double v) { out[i] = v; }, flag);
    }
}

// B cut into planes[3][KSP][NT][64][8] bf16, zero outside the matrix: element j of lane l of (ks, tile) is B(k = 16 ks + 8 (l >> 5)
// + j, n = 32 tile + (l & 31)). BLAY 0: B is [K][N] (ldb), 1: B is stored [N][K] (ldb). A thread per (fragment, half of it).
template <int BLAY>
__global__ __launch_bounds__(256) void gemm_rsplit_planes_kernel(const float* __restrict__ B, int N, int K, int ldb, int KSP, int NT,
                                                                 unsigned char* __restrict__ planes) {
    const int total = KSP * NT * 64 * 2;
    const size_t plane_stride = static_cast<size_t>(KSP) * NT * 1024;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        // BLAY 0: consecutive threads walk n (rows of B are contiguous in n); BLAY 1: consecutive threads walk k
        int ks, tile, l, half;
        if (BLAY == 0) { const int l31 = idx & 31; int r = idx >> 5; tile = r % NT; r /= NT; half = r & 1; r >>= 1; const int lk = r & 1; ks = r >> 1; l = l31 + 32 * lk; }
        else { half = idx & 1; int r = idx >> 1; const int lk = r & 1; r >>= 1; ks = r % KSP; r /= KSP; const int l31 = r & 31; tile = r >> 5; l = l31 + 32 * lk; }
        if (ks >= KSP || tile >= NT) continue;
        const int n = 32 * tile + (l & 31), k0 = 16 * ks + 8 * (l >> 5) + 4 * half;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = n < N && k0 + e < K;
            const size_t off = ok ? (BLAY == 0 ? static_cast<size_t>(k0 + e) * ldb + n : static_cast<size_t>(n) * ldb + k0 + e) : 0;
            const float v = B[off];
            x[e] = ok ? v : 0.f;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        rs_pair(x[0], x[1], h0, m0, l0);
        rs_pair(x[2], x[3], h1, m1, l1);
        unsigned char* p = planes + ((static_cast<size_t>(ks) * NT + tile) * 64 + l) * 16 + half * 8;
        *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(p + 2 * plane_stride) = make_uint2(l0, l1);
    }
}

struct RsPlan { int waves, rch, ksp, nt; size_t lds; };

// the shapes the kernel covers: N <= 320, N and K multiples of 4, the panel's planes in LDS
// gather_window > 0: the forward product with the word gather-mean inside (GATH): at most five float4s of the panel per thread
// (five word rows in flight for each), at most eight waves
bool rs_plan(int b_layout, int M, int N, int K, bool colstats, bool rowsq, bool bn, RsPlan* p, int gather_window = 0) {
    if (M <= 0 || N < 32 || K < 16 || N > 320 || (N % 4) || (K % 4)) return false;
    if ((colstats && rowsq) || (b_layout == 0 && (bn || rowsq)) || (bn && (b_layout != 1 || colstats))) return false;
    const int nt = (N + 31) / 32;
    const int waves = nt <= 4 ? 4 : (nt <= 8 ? 8 : 10);
    int ks = (K + 15) / 16;
    const int ksp = ks + (ks & 1);
    const int threads = 64 * waves, rounds = (SM * 4 * ksp + threads - 1) / threads;
    const int rch = rounds <= 4 ? 4 : (rounds == 5 ? 5 : (rounds <= 8 ? 8 : 0));
    if (!rch) return false;
    if (bn && rch > 4) return false;            // (the instantiations: PRE comes with four float4s + four of `pre` per thread)
    if (gather_window && (bn || b_layout != 0 || rch > 5 || waves > 8 || gather_window > kRsGatherMaxWindow)) return false;
    const int np = 32 * waves;
    const size_t img = std::max<size_t>(static_cast<size_t>(3) * ksp * 1024, static_cast<size_t>(SM) * (np + 4) * 4);
    const size_t lds = img + (static_cast<size_t>(2) * np + (bn ? 4 * static_cast<size_t>(K) : 0) + 4 +
                              (gather_window ? 3 * static_cast<size_t>(SM) * gather_window + kLazyHistory : 0)) * sizeof(float);
    if (lds > kRsLdsMax) return false;
    p->waves = waves; p->rch = rch; p->ksp = ksp; p->nt = nt; p->lds = lds;
    return true;
}

template <bool PRE, int WAVES, int RCH, int NPROD, int GATH = 0>
bool rs_launch(const RsArgs& g, int grid, size_t lds, hipStream_t s, const typename RsGatherArgs<GATH>::type& ga = typename RsGatherArgs<GATH>::type{}) {
    // more than 64 KB of dynamic LDS is an opt-in per kernel and per DEVICE
    static std::atomic<bool> attr_set[kRsMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kRsMaxDevices) return false;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rsplit_kernel<PRE, WAVES, RCH, NPROD, GATH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kRsLdsMax)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev].store(true, std::memory_order_release);
    }
    NVSM_LAUNCH((gemm_rsplit_kernel<PRE, WAVES, RCH, NPROD, GATH>), dim3(grid), dim3(64 * WAVES), lds, s, g, ga);
    return true;
}

}  // namespace

size_t gemm_rsplit_planes_bytes(int N, int K) {
    const int ks = (K + 15) / 16, ksp = ks + (ks & 1), nt = (N + 31) / 32;
    return static_cast<size_t>(3) * ksp * nt * 1024;
}

PlaneTarget gemm_rsplit_plane_target(int N, int K, void* planes, int transposed) {
    const int ks = (K + 15) / 16, ksp = ks + (ks & 1), nt = (N + 31) / 32;
    return PlaneTarget{static_cast<unsigned char*>(planes), static_cast<size_t>(ksp) * nt * 1024, 2, nt, transposed};
}
void launch_gemm_rsplit_planes(int b_layout, const float* B, int N, int K, int ldb, void* planes, hipStream_t s) {
    const int ks = (K + 15) / 16, ksp = ks + (ks & 1), nt = (N + 31) / 32;
    const int total = ksp * nt * 128, grid = (total + 255) / 256;
    if (b_layout == 0) NVSM_LAUNCH((gemm_rsplit_planes_kernel<0>), dim3(grid), dim3(256), 0, s, B, N, K, ldb, ksp, nt, static_cast<unsigned char*>(planes));
    else NVSM_LAUNCH((gemm_rsplit_planes_kernel<1>), dim3(grid), dim3(256), 0, s, B, N, K, ldb, ksp, nt, static_cast<unsigned char*>(planes));
}

bool gemm_rsplit_covers(int b_layout, int M, int N, int K, bool colstats, bool rowsq, bool bn) {
    RsPlan p;
    return gemm_split_products() != 0 && tuning().gemm_rsplit && rs_plan(b_layout, M, N, K, colstats, rowsq, bn, &p);
}
bool gemm_rsplit_gather_covers(int M, int N, int K, bool colstats, int window) {
    RsPlan p;
    return window >= 1 && gemm_split_products() != 0 && tuning().gemm_rsplit && rs_plan(0, M, N, K, colstats, false, false, &p, window);
}

// true: launched. A [M][K] row-major, 16 B aligned operands, leading dimensions multiples of 4. ws: the planes of B in this
// kernel's layout (GemmSplitWs::rplanes; cut here, on `s`, unless ws->rready says they are current).
bool launch_gemm_rsplit(int b_layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                        float alpha, const float* bias_n, hipStream_t s, double* colstats, const GridSumWs* sums, float* rowsq,
                        float rowsq_scale, GemmSplitWs* ws, const BnDxFused* bn, const GatherFused* gf) {
    const int nprod = gemm_split_products();
    if (!nprod || !tuning().gemm_rsplit || !ws || !ws->rplanes || ws->rbytes < gemm_rsplit_planes_bytes(N, K)) return false;
    if ((lda % 4) || (ldb % 4) || (ldc % 4) || lda < K) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) % 16) return false;
    const bool fused_bn = bn && bn->pre;
    if (fused_bn && (bn->dy != A || reinterpret_cast<uintptr_t>(bn->pre) % 16)) return false;
    RsPlan p;
    if (gf && (gf->window < 1 || lda != K || !gf->table || !gf->idx || reinterpret_cast<uintptr_t>(gf->table) % 16)) return false;
    if (!rs_plan(b_layout, M, N, K, colstats != nullptr, rowsq != nullptr, fused_bn, &p, gf ? gf->window : 0)) return false;
    const int grid = (M + SM - 1) / SM;
    RsArgs g{};
    g.A = A; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.KSP = p.ksp; g.NT = p.nt;
    g.alpha = alpha; g.bias_n = bias_n; g.colstats = colstats; g.rowsq = rowsq; g.rowsq_scale = rowsq_scale;
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
    if (!ws->rready) { launch_gemm_rsplit_planes(b_layout, B, N, K, ldb, ws->rplanes, s); ws->rready = true; }
    g.planes = static_cast<const unsigned char*>(ws->rplanes);
    if (gf) {      // the word gather-mean inside (forward product): A is written — the phrase matrix — not read
        const bool lz = gf->lazy && gf->lazy->stamp;
        RsGather ge{gf->table, gf->idx, gf->wts, gf->window, const_cast<float*>(A)};
        RsGatherLazy gl{gf->table, gf->idx, gf->wts, gf->window, const_cast<float*>(A), lz ? *gf->lazy : LazyView{}};
#define NVSM_RSG_CASE(W, R)                                                                                             \
        if (p.waves == W && p.rch == R) {                                                                               \
            if (lz) return nprod == 9 ? rs_launch<false, W, R, 9, 2>(g, grid, p.lds, s, gl) : rs_launch<false, W, R, 6, 2>(g, grid, p.lds, s, gl); \
            return nprod == 9 ? rs_launch<false, W, R, 9, 1>(g, grid, p.lds, s, ge) : rs_launch<false, W, R, 6, 1>(g, grid, p.lds, s, ge); \
        }
        NVSM_RSG_CASE(4, 4) NVSM_RSG_CASE(4, 5) NVSM_RSG_CASE(8, 4) NVSM_RSG_CASE(8, 5)
#undef NVSM_RSG_CASE
        return false;
    }
#define NVSM_RS_CASE(W, R)                                                                                              \
    if (p.waves == W && p.rch == R) {                                                                                   \
        if (fused_bn) { if constexpr (R == 4) return nprod == 9 ? rs_launch<true, W, R, 9>(g, grid, p.lds, s) : rs_launch<true, W, R, 6>(g, grid, p.lds, s); else return false; } \
        return nprod == 9 ? rs_launch<false, W, R, 9>(g, grid, p.lds, s) : rs_launch<false, W, R, 6>(g, grid, p.lds, s); \
    }
    NVSM_RS_CASE(4, 4) NVSM_RS_CASE(4, 5) NVSM_RS_CASE(4, 8) NVSM_RS_CASE(8, 4) NVSM_RS_CASE(8, 5) NVSM_RS_CASE(8, 8)
    NVSM_RS_CASE(10, 4) NVSM_RS_CASE(10, 5) NVSM_RS_CASE(10, 8)
#undef NVSM_RS_CASE
    return false;
}

}  // namespace cunvsm
