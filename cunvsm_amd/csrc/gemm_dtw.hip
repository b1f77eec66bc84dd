// dT[d_w][d_e] = phraseᵀ[d_w x B] · dx[B x d_e] (cpp/params.cu:526-531) at PER-RANK batch sizes (the 6 400 windows a rank of the 8-GPU
// job holds, the LSE recipe's 4 096), in workgroups of ONE wave. Round 6.
//
// At these batch sizes the product does not run alone: it sits on side stream 2 next to both table passes (thousands of short-lived
// waves at five per SIMD, 465 of a SIMD's 512 registers taken), and the chain it heads — product → projection update — is what the
// next step's forward product waits for (profiles/NOTES_r06.md §1: without the product the per-rank steps are 4-6 % shorter). The
// exact-fp32 tiled kernel that ran here (gemm_f32_mfma_kernel<1, 0>: 256 threads, 32 KB of LDS, ~130 registers a wave) takes 38 us
// alone and 106-124 us in the step: a workgroup of four waves needs room on all four SIMDs of a CU at once and gets it only when
// several pass waves retire together; gemm_dt.hip's whole-CU workgroups fare worse (NOTES_r05.md §1). So this kernel asks for the
// least a dispatch can ask for — one wave, no LDS, no barrier, under a hundred registers — and goes wherever ONE pass wave has left:
//   * a wave owns a 32 x 64 tile of the output (phrase columns x dx columns) over one slab of the batch (split-K; the slabs are
//     added by the projection update, model.cpp fuse_slab_sum_, or by launch_splitk_reduce);
//   * both operands enter with the batch as the reduction dimension, i.e. transposed against the way they lie in memory. There is
//     nothing to transpose when every lane fetches its OWN fragment elements: lane (x, kq) of v_mfma_f32_32x32x16_bf16 holds eight
//     consecutive k of column x, which are eight dword loads a row pitch apart — 32 lanes wide each (128 B rows: whole sectors), the
//     operands (14 MB at batch 6 400) L2 / Infinity-Cache resident, the next k step's loads in flight while this one is cut;
//   * the arithmetic is gemm_split.hip's: every fp32 element cut exactly into three bf16 pieces (x = h + m + l), six (or nine) bf16
//     MFMAs per product with fp32 accumulation, smallest products first, the two column blocks' accumulators in rotation.
// What it costs is VALU — an element is cut by every wave that uses it (phrase x 4, dx x 10): 160 VALU per 12 MFMAs — which is why the
// large-batch kernels stage through LDS; here the matrix pipe is idle anyway and the waves are few (640 at batch 6 400).
#include "kernels.h"
#include "device_utils.h"

namespace cunvsm {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWM = 32, kWN = 64, kWK = 16;      // tile of a wave: 32 phrase columns x 64 dx columns; k step = 16 batch rows

struct DtwArgs {
    const float* A; const float* B;             // phrase [rows][M] (lda) / dx [rows][N] (ldb)
    int lda, ldb;
    float* P; size_t p_stride; int ldc;         // partial [slabs][M][N]
    int rows, M, N, slab_rows, slabs, tiles_m, tiles_n;
};

// a - b as ONE v_sub_f32 (gemm_split.hip split_sub: packed fp32 VALU is slow next to MFMAs)
__device__ __forceinline__ float dtw_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x0, x1 (consecutive k) -> one 32-bit word per plane, x0's piece in the lower half: h = bf16(x), m = bf16(x - h), l = x - h - m
__device__ __forceinline__ void dtw_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
    const float r0 = dtw_sub(x0, __uint_as_float(h << 16)), r1 = dtw_sub(x1, __uint_as_float(h & 0xffff0000u));
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
    const float s0 = dtw_sub(r0, __uint_as_float(m << 16)), s1 = dtw_sub(r1, __uint_as_float(m & 0xffff0000u));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}
struct DtwFrag { u32x4 h, m, l; };
__device__ __forceinline__ void dtw_cut(const float (&x)[8], bool live, DtwFrag& f) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        unsigned h, m, l;
        dtw_pair(live ? x[2 * r] : 0.f, live ? x[2 * r + 1] : 0.f, h, m, l);
        f.h[r] = h; f.m[r] = m; f.l[r] = l;
    }
}
__device__ __forceinline__ f32x16 dtw_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int NPROD>
__global__ __launch_bounds__(64) void gemm_dtw_kernel(DtwArgs g) {
    const int lane = threadIdx.x, x = lane & 31, kq = lane >> 5;
    // block -> (slab, tile): the tiles of a slab are neighbours (they read the same rows of both operands)
    const int tiles = g.tiles_m * g.tiles_n;
    const int slab = blockIdx.x / tiles, tile = blockIdx.x - slab * tiles;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int m0 = tm * kWM, n0 = tn * kWN;
    const int row_begin = slab * g.slab_rows;
    const int row_end = (row_begin + g.slab_rows) < g.rows ? (row_begin + g.slab_rows) : g.rows;
    const int T = (row_end - row_begin + kWK - 1) / kWK;
    // this lane's column of each operand block (past the matrix: the last column, multiplied as zeros)
    const bool a_ok = m0 + x < g.M, b0_ok = n0 + x < g.N, b1_ok = n0 + 32 + x < g.N;
    const float* pa = g.A + (a_ok ? m0 + x : g.M - 1);
    const float* pb0 = g.B + (b0_ok ? n0 + x : g.N - 1);
    const float* pb1 = g.B + (b1_ok ? n0 + 32 + x : g.N - 1);

    // k step t: rows row_begin + 16 t + 8 kq + j, j = 0 .. 7 (rows past the slab: its last row again, cut as zeros)
    auto load = [&](int t, float (&a)[8], float (&b0)[8], float (&b1)[8]) {
        const int r0 = row_begin + kWK * t + 8 * kq;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int r = r0 + j;
            r = r < row_end ? r : row_end - 1;
            a[j] = pa[static_cast<size_t>(r) * g.lda];
            b0[j] = pb0[static_cast<size_t>(r) * g.ldb];
            b1[j] = pb1[static_cast<size_t>(r) * g.ldb];
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    float a[8], b0[8], b1[8], an[8], b0n[8], b1n[8];
    if (T > 0) load(0, a, b0, b1);
    for (int t = 0; t < T; ++t) {
        // the next k step's elements are requested before this one's are cut (past the last step: the last one again, unused)
        load(t + 1 < T ? t + 1 : t, an, b0n, b1n);
        // rows of this k step past the slab's end contribute zeros: all eight of a lane are in or out together only when the slab
        // length is a multiple of 8 — in general the tail is cut per element
        const int r0 = row_begin + kWK * t + 8 * kq;
        DtwFrag fa, fb0, fb1;
        if (r0 + 8 <= row_end) {
            dtw_cut(a, a_ok, fa); dtw_cut(b0, b0_ok, fb0); dtw_cut(b1, b1_ok, fb1);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = r0 + j < row_end;
                a[j] = in ? a[j] : 0.f; b0[j] = in ? b0[j] : 0.f; b1[j] = in ? b1[j] : 0.f;
            }
            dtw_cut(a, a_ok, fa); dtw_cut(b0, b0_ok, fb0); dtw_cut(b1, b1_ok, fb1);
        }
        // smallest products first; operands fed swapped (the accumulator's rows run along the dx columns: 16 B stores); the two
        // column blocks in rotation (a chain of dependent MFMAs runs at the pipe's latency, not its rate)
        if (NPROD == 9) {
            acc0 = dtw_mfma(fb0.l, fa.l, acc0); acc1 = dtw_mfma(fb1.l, fa.l, acc1);
            acc0 = dtw_mfma(fb0.l, fa.m, acc0); acc1 = dtw_mfma(fb1.l, fa.m, acc1);
            acc0 = dtw_mfma(fb0.m, fa.l, acc0); acc1 = dtw_mfma(fb1.m, fa.l, acc1);
        }
        acc0 = dtw_mfma(fb0.l, fa.h, acc0); acc1 = dtw_mfma(fb1.l, fa.h, acc1);
        acc0 = dtw_mfma(fb0.h, fa.l, acc0); acc1 = dtw_mfma(fb1.h, fa.l, acc1);
        acc0 = dtw_mfma(fb0.m, fa.m, acc0); acc1 = dtw_mfma(fb1.m, fa.m, acc1);
        acc0 = dtw_mfma(fb0.m, fa.h, acc0); acc1 = dtw_mfma(fb1.m, fa.h, acc1);
        acc0 = dtw_mfma(fb0.h, fa.m, acc0); acc1 = dtw_mfma(fb1.h, fa.m, acc1);
        acc0 = dtw_mfma(fb0.h, fa.h, acc0); acc1 = dtw_mfma(fb1.h, fa.h, acc1);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = an[j]; b0[j] = b0n[j]; b1[j] = b1n[j]; }
    }

    // ---- this slab's partial: acc_c[4 q + r] = P[slab][m0 + x][n0 + 32 c + 8 q + 4 kq + r] ----
    float* P = g.P + static_cast<size_t>(slab) * g.p_stride;
    const int m = m0 + x;
    if (m < g.M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + 8 * q + 4 * kq;
            if (n < g.N) *reinterpret_cast<f32x4*>(P + static_cast<size_t>(m) * g.ldc + n) = f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
            if (n + 32 < g.N) *reinterpret_cast<f32x4*>(P + static_cast<size_t>(m) * g.ldc + n + 32) = f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
        }
    }
}

int dtw_slab_rows(int rows, int want) {
    if (want < 1) want = 1;
    const int len = (rows + want - 1) / want;
    return ((len + kWK - 1) / kWK) * kWK;
}

}  // namespace

// shapes the wave-sized kernel takes: columns in whole float4s of the output (N % 4 == 0)
bool gemm_dtw_covers(int M, int N, int rows) {
    return gemm_split_products() != 0 && M >= 1 && N >= 4 && N % 4 == 0 && rows >= 1 && M <= 4096 && N <= 4096;
}
// slabs the product will use for `want` (slab lengths are multiples of 16 rows): never more than `want`
int gemm_dtw_slabs(int rows, int want) {
    const int len = dtw_slab_rows(rows, want);
    return (rows + len - 1) / len;
}

// A [rows][M] (lda), B [rows][N] (ldb); partial [gemm_dtw_slabs(rows, want)][M][N] (ldc = N; one slab: the product itself).
// false: shape not covered, nothing launched.
bool launch_gemm_dtw(const float* A, const float* B, float* partial, int M, int N, int rows, int lda, int ldb, int want_slabs, hipStream_t s) {
    if (!gemm_dtw_covers(M, N, rows) || lda < M || ldb < N) return false;
    if (reinterpret_cast<uintptr_t>(partial) % 16) return false;
    DtwArgs g{};
    g.A = A; g.B = B; g.lda = lda; g.ldb = ldb;
    g.P = partial; g.p_stride = static_cast<size_t>(M) * N; g.ldc = N;
    g.rows = rows; g.M = M; g.N = N; g.slab_rows = dtw_slab_rows(rows, want_slabs);
    g.slabs = gemm_dtw_slabs(rows, want_slabs);
    g.tiles_m = (M + kWM - 1) / kWM; g.tiles_n = (N + kWN - 1) / kWN;
    const int64_t grid = static_cast<int64_t>(g.slabs) * g.tiles_m * g.tiles_n;
    if (grid > (int64_t(1) << 30)) return false;
    if (gemm_split_products() == 9) NVSM_LAUNCH((gemm_dtw_kernel<9>), dim3(static_cast<unsigned>(grid)), dim3(64), 0, s, g);
    else NVSM_LAUNCH((gemm_dtw_kernel<6>), dim3(static_cast<unsigned>(grid)), dim3(64), 0, s, g);
    return true;
}

}  // namespace cunvsm
