// dT[d_w][d_e] = phraseᵀ[d_w x B] · dx[B x d_e], the projection gradient (cpp/params.cu:526-531), on the bf16 matrix pipe at
// fp32 accuracy — the split-bf16 arithmetic of gemm_split.hip (every fp32 operand cut exactly into three bf16 pieces, six or
// nine bf16 MFMAs per product, fp32 accumulation) for the product whose LONG dimension is the reduction: split-K over the batch.
// A workgroup (eight waves) owns a slab of the batch and one half of the d_e columns:
//   * K runs in tiles of 32 batch rows. 448 of the 512 threads each fetch an 8-row x 4-column piece of the tile (phrase:
//     80 column groups x 4 row octets incl. the padding to 320 columns; dx half: 32 x 4) a turn ahead, cut it, and store —
//     per column and plane — the eight consecutive-row bf16 values as ONE 16 B LDS word: exactly the fragment of an MFMA
//     whose K runs along the batch (the transposition is free: it happens in registers). Column pitch 64 B with the octets
//     rotated per column group so that fragment reads are conflict-free (dt_slot). One LDS image (86 KB), two barriers per
//     tile with only the twelve LDS stores between them (the cutting is pinned between the MFMAs of the tile before);
//   * wave (cp, rh) multiplies the 32-column block cp of the half against the 32-row blocks 5 rh ... 5 rh + 4 of d_w with
//     v_mfma_f32_32x32x16_bf16 (two k steps per tile), waves w and w + 4 share a SIMD; 80 accumulator registers;
//   * the slab's partial product goes to partial[slab] and launch_splitk_reduce adds the slabs in order, as before.
// What bounds it (tools/exp/dt_ticks.py, tools/exp/dt_alone.sh): instruction issue. Both operands are batch-sized, so every
// tile brings 17.8 k elements to cut (≈ 900 non-MFMA instructions per SIMD and turn against 120 MFMAs, of which the matrix
// pipe's shadow hides about five each): a turn takes the MFMAs' time PLUS most of the staging's, 4.4 us per tile and
// workgroup against 1.8 us of MFMAs — the same with 16x16x32 or 32x32x16 MFMAs, with or without LDS bank conflicts, with
// the cutting between the MFMAs or behind the barrier, with the loads one or two turns ahead (all measured). 64 slabs:
// 110 us alone (128: 70 us; the tiled exact-fp32 kernel: 256 us at 16 slabs, 153 us at 50).
// Where it runs: model.cpp step() — on the MAIN stream for large batches of eager tables (its workgroups take a CU's LDS and
// most of its registers and starve behind the table passes' thousands of small workgroups on a side stream).
#include "kernels.h"
#include "device_utils.h"

#include <atomic>
#include <cstdlib>

namespace cunvsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kDtWaves = 8, kDtThreads = kDtWaves * 64;
constexpr int kDtPitch = 64;                                   // bytes per column of a plane: 32 bf16
// Octet o (eight consecutive rows = 16 B) of column c sits at slot (o + c / 4) mod 4 of the column's 64 B: ds_read_b128
// is serviced in four groups of sixteen lanes that mix two octets — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... —
// (MI355X_MICROARCH.md, LDS); a fragment of the 32x32x16 MFMA is column (lane & 31), octet 2 ks + (lane >> 5), so a group holds
// sixteen columns of one octet, four of them per residue of c mod 4 (which share the 64 B span), and this rotation gives those
// four different slots: every group falls on sixteen different bank quads. (A padded pitch of 80 B, conflict-free for sixteen CONSECUTIVE lanes, left three of the eight lanes of the second
// octet on the banks of the first: every fragment read took twice its cycles, and the reads — 288 KB per tile and CU —
// were as long as the MFMAs.)
__device__ __forceinline__ int dt_slot(int c, int o) { return (o + (c >> 2)) & 3; }
constexpr int kDtMaxM = 320, kDtHalfN = 128;
constexpr int kDtRB = 5;                                       // 32-row blocks per wave (two waves cover up to 10 = 320 rows)
constexpr int kDtMaxDevices = 64;

struct DtArgs {
    const float* A; const float* B; float* P;
    int rows, M, N, lda, ldb, ldc;       // rows = batch (the reduction); A [rows][M] (lda), B [rows][N] (ldb), P [slabs][M][N] (ldc)
    int slab_rows;                       // batch rows per slab, a multiple of 32
    size_t p_stride;
    int mpad;                            // 32 · row blocks
};

__device__ __forceinline__ void dt_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    // round-to-nearest pieces (v_cvt_pk_bf16_f32: two elements per instruction): h = bf16(x), m = bf16(x - h), l = x - h - m.
    // Both differences are exact in fp32 and the last one has at most eight significant bits, so x = h + m + l exactly;
    // the pieces are at most half an ulp of the piece above them (a cut by truncation leaves up to a whole one, all of x's
    // sign): the partial products a six-product run leaves out are below 2^-26 of a·b and of either sign.
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r = v - f32x2_t{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    const f32x2_t s2 = r - f32x2_t{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(s2, bf16x2_t));
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
// 32x32x16: half as many instructions as 16x16x32 for the same flops, and 32 cycles of matrix pipe behind each in which
// the cutting, the LDS traffic and the address arithmetic of the staging issue (with the 16-cycle instruction the kernel
// took MFMA time + staging time: 10 k cycles per tile for 4.4 k of MFMAs)
__device__ __forceinline__ f32x16 dt_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifdef NVSM_DT_TIMING
// experiments (make dbg; tools/exp/dt_ticks.py): shader-clock stamps of waves 0 and 4 of workgroup 0 inside the K loop
__device__ unsigned long long g_dt_ticks[2 * 16 * 8];
#define DT_TICK(kt, point) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (kt) < 16) \
    g_dt_ticks[(threadIdx.x >> 8) * 128 + (kt) * 8 + (point)] = clock64(); } while (0)
#else
#define DT_TICK(kt, point) do {} while (0)
#endif

template <int NPROD>
__global__ __launch_bounds__(kDtThreads) void gemm_dt_kernel(DtArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dt_lds[];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, kh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cp = w & 3, rh = w >> 2;                 // this wave: 32-column block cp of the half, row blocks 5 rh ... 5 rh + 4
    const int half = blockIdx.x & 1, slab = blockIdx.x >> 1;
    const int n0 = half * kDtHalfN;
    const int row_begin = slab * g.slab_rows;
    const int row_end = (row_begin + g.slab_rows) < g.rows ? (row_begin + g.slab_rows) : g.rows;
    const int KT = (row_end - row_begin + 31) / 32;
    const int rblocks = g.mpad / 32;
    const int rb0 = rh * kDtRB;
    const int nrb = (rblocks - rb0) < kDtRB ? (rblocks - rb0 > 0 ? rblocks - rb0 : 0) : kDtRB;
    const int a_plane = g.mpad * kDtPitch;                                  // bytes of one plane of the phrase part
    constexpr int b_plane = kDtHalfN * kDtPitch;
    unsigned char* a_img = dt_lds;                                          // [3][mpad][64]
    unsigned char* b_img = dt_lds + 3 * a_plane;                            // [3][128][64]

    // ---- this thread's staging task: 8 rows (octet o of the tile) x 4 columns (group cg) of A, or of this half of B ----
    const int a_groups = g.mpad / 4;                                        // column groups of A incl. the padding (zeros)
    const int a_tasks = 4 * a_groups;
    const bool is_a = tid < a_tasks;
    const bool is_b = !is_a && tid < a_tasks + 4 * (kDtHalfN / 4);
    const int tt = is_a ? tid : tid - a_tasks;
    const int o = tt & 3, cg = tt >> 2;
    const bool col_ok = is_a ? (4 * cg < g.M) : (is_b && n0 + 4 * cg < g.N);       // M % 4 == 0, N % 4 == 0
    // (threads without a piece, and pieces in the padding columns, load column 0 of A: never stored / stored as zeros)
    const float* src = (is_b ? g.B : g.A) + (col_ok ? (is_a ? 4 * cg : n0 + 4 * cg) : 0);
    const int ld = is_b ? g.ldb : g.lda;
    unsigned char* dst = (is_a ? a_img : b_img) + (4 * cg) * kDtPitch + dt_slot(4 * cg, o) * 16;      // (columns 4 cg .. 4 cg + 3 share c / 4)
    const int dplane = is_a ? a_plane : b_plane;

    // two register sets: `raw` holds the piece of the tile that is cut during this turn, `nxt` receives the one after it — a
    // load issued at the top of a turn has the whole turn (2 us) to arrive; issued at its bottom it was waited for 0.5 us later
    u32x4 raw[8], nxt[8];
    auto load_tile = [&](int kt, u32x4 (&dstr)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int row = row_begin + 32 * kt + 8 * o + r;
            row = row < row_end ? row : row_end - 1;                        // (clamped: zeroed in store_tile)
            dstr[r] = *reinterpret_cast<const u32x4*>(src + static_cast<size_t>(row) * ld);
        }
    };
    // the piece of tile kt in `raw` cut into fragment words (column e: its eight rows, one 16 B word per plane) — register work,
    // done while the other waves still multiply — and, behind the barrier, the twelve LDS stores
    u32x4 ch[4], cm[4], cl[4];
    auto cut_col = [&](int kt, int e) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ra = row_begin + 32 * kt + 8 * o + 2 * j;
            const float x0 = (col_ok && ra < row_end) ? __uint_as_float(raw[2 * j][e]) : 0.f;
            const float x1 = (col_ok && ra + 1 < row_end) ? __uint_as_float(raw[2 * j + 1][e]) : 0.f;
            unsigned hh, mm, ll;
            dt_split_pair(x0, x1, hh, mm, ll);
            ch[e][j] = hh; cm[e][j] = mm; cl[e][j] = ll;
        }
    };
    auto cut_tile = [&](int kt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cut_col(kt, e);
    };
    auto write_tile = [&] {
        if (!(is_a || is_b)) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned char* p = dst + e * kDtPitch;
            *reinterpret_cast<u32x4*>(p) = ch[e];
            *reinterpret_cast<u32x4*>(p + dplane) = cm[e];
            *reinterpret_cast<u32x4*>(p + 2 * dplane) = cl[e];
        }
    };

    f32x16 acc[kDtRB];
#pragma unroll
    for (int rb = 0; rb < kDtRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

    if (KT > 0) {
        load_tile(0, raw);
        cut_tile(0);
        write_tile();
        load_tile(KT > 1 ? 1 : 0, raw);
    }
    __syncthreads();
    // fragment of column c, k step ks: the 16 B word of octet 2 ks + kh
    auto frag = [&](const unsigned char* img, int plane_bytes, int col, int ks, u32x4& h, u32x4& m, u32x4& l) {
        const unsigned char* p = img + col * kDtPitch + dt_slot(col, 2 * ks + kh) * 16;
        h = *reinterpret_cast<const u32x4*>(p);
        m = *reinterpret_cast<const u32x4*>(p + plane_bytes);
        l = *reinterpret_cast<const u32x4*>(p + 2 * plane_bytes);
    };
    for (int kt = 0; kt < KT; ++kt) {
        DT_TICK(kt, 0);
        load_tile(kt + 2 < KT ? kt + 2 : KT - 1, nxt);     // (past the last tile: a harmless repeat, the same loads every turn)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 1) DT_TICK(kt, 1);
            u32x4 bh, bm, bl;
            frag(b_img, b_plane, 32 * cp + i, ks, bh, bm, bl);
            u32x4 ah, am, al;
            frag(a_img, a_plane, 32 * rb0 + i, ks, ah, am, al);
#pragma unroll
            for (int rb = 0; rb < kDtRB; ++rb) {
                // the fragments of row block rb + 1 are read while block rb is multiplied (a wave with fewer blocks re-reads its first)
                u32x4 nh = ah, nm = am, nl = al;
                if (rb + 1 < kDtRB) frag(a_img, a_plane, 32 * (rb0 + (rb + 1 < nrb ? rb + 1 : 0)) + i, ks, nh, nm, nl);
                __builtin_amdgcn_sched_barrier(0);
                // the next tile's piece is cut between the MFMAs (register work only): column e in k step e / 2, block 1 + 2 (e & 1)
                if (rb == 1 || rb == 3) {
                    const int e = 2 * ks + (rb >> 1);
                    cut_col(kt + 1, e);
                    // (pins the cutting here: its results are only stored behind the barrier, and left alone the compiler sinks
                    //  the whole computation to that store — measured: the turn took MFMA time + cutting time)
                    asm volatile("" : "+v"(ch[e]), "+v"(cm[e]), "+v"(cl[e]));
                }
                if (rb < nrb) {
                    if (NPROD == 9) {
                        acc[rb] = dt_mfma(bl, al, acc[rb]);
                        acc[rb] = dt_mfma(bl, am, acc[rb]);
                        acc[rb] = dt_mfma(bm, al, acc[rb]);
                    }
                    acc[rb] = dt_mfma(bl, ah, acc[rb]);
                    acc[rb] = dt_mfma(bh, al, acc[rb]);
                    acc[rb] = dt_mfma(bm, am, acc[rb]);
                    acc[rb] = dt_mfma(bm, ah, acc[rb]);
                    acc[rb] = dt_mfma(bh, am, acc[rb]);
                    acc[rb] = dt_mfma(bh, ah, acc[rb]);
                }
                __builtin_amdgcn_sched_barrier(0);
                ah = nh; am = nm; al = nl;
            }
        }
        // (the piece of tile kt + 1 has been cut on the way; past the last tile: zeros, written to an image nobody reads)
        DT_TICK(kt, 2);
        __syncthreads();                                   // everybody has read tile kt
        DT_TICK(kt, 3);
        write_tile();
        DT_TICK(kt, 4);
#pragma unroll
        for (int r = 0; r < 8; ++r) raw[r] = nxt[r];
        DT_TICK(kt, 5);
        __syncthreads();
        DT_TICK(kt, 6);
    }

    // ---- this slab's partial: acc[rb][4 g + r] = P[slab][32 (rb0 + rb) + i][n0 + 32 cp + 8 g + 4 kh + r] ----
    // (operands fed swapped: the accumulator's rows run along the columns of dx, four consecutive ones per register quad)
    float* P = g.P + static_cast<size_t>(slab) * g.p_stride;
#pragma unroll
    for (int rb = 0; rb < kDtRB; ++rb) {
        if (rb < nrb) {
            const int m = 32 * (rb0 + rb) + i;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + 32 * cp + 8 * gq + 4 * kh;
                if (m < g.M && n < g.N)
                    *reinterpret_cast<f32x4*>(P + static_cast<size_t>(m) * g.ldc + n) =
                        f32x4{acc[rb][4 * gq], acc[rb][4 * gq + 1], acc[rb][4 * gq + 2], acc[rb][4 * gq + 3]};
            }
        }
    }
}

size_t gemm_dt_lds_bytes(int M) { return static_cast<size_t>(3) * (32 * ((M + 31) / 32) + kDtHalfN) * kDtPitch; }

bool gemm_dt_covers(int M, int N, int rows) {
    return gemm_split_products() != 0 && M % 4 == 0 && N % 4 == 0 && M >= 16 && M <= kDtMaxM && N > kDtHalfN && N <= 2 * kDtHalfN && rows >= 64;
}

// slabs the product will use for `want` (slab lengths are multiples of 32 rows)
int gemm_dt_slabs(int rows, int want) {
    if (want < 1) want = 1;
    int len = (rows + want - 1) / want;
    len = ((len + 31) / 32) * 32;
    return (rows + len - 1) / len;
}

// partial [slabs][M][N] (ldc = N); the caller adds the slabs (launch_splitk_reduce). false: shape not covered, nothing launched.
bool launch_gemm_dt(const float* A, const float* B, float* partial, int M, int N, int rows, int lda, int ldb, int want_slabs,
                    hipStream_t s) {
    if (!gemm_dt_covers(M, N, rows) || (lda % 4) || (ldb % 4)) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(partial)) % 16) return false;
    const int nprod = gemm_split_products();
    DtArgs g{};
    g.A = A; g.B = B; g.P = partial; g.rows = rows; g.M = M; g.N = N; g.lda = lda; g.ldb = ldb; g.ldc = N;
    const int slabs = gemm_dt_slabs(rows, want_slabs);
    int len = (rows + (want_slabs < 1 ? 1 : want_slabs) - 1) / (want_slabs < 1 ? 1 : want_slabs);
    g.slab_rows = ((len + 31) / 32) * 32;
    g.p_stride = static_cast<size_t>(M) * N;
    g.mpad = 32 * ((M + 31) / 32);
    const size_t lds = gemm_dt_lds_bytes(M);
    static std::atomic<bool> attr_set[kDtMaxDevices][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDtMaxDevices) return false;
    const int which = nprod == 9 ? 1 : 0;
    if (!attr_set[dev][which].load(std::memory_order_acquire)) {
        const void* fn = which ? reinterpret_cast<const void*>(&gemm_dt_kernel<9>) : reinterpret_cast<const void*>(&gemm_dt_kernel<6>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (kDtMaxM + kDtHalfN) * kDtPitch) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev][which].store(true, std::memory_order_release);
    }
    (void)hipGetLastError();
    if (which) NVSM_LAUNCH((gemm_dt_kernel<9>), dim3(2 * slabs), dim3(kDtThreads), lds, s, g);
    else NVSM_LAUNCH((gemm_dt_kernel<6>), dim3(2 * slabs), dim3(kDtThreads), lds, s, g);
    return hipGetLastError() == hipSuccess;
}

}  // namespace cunvsm

#ifdef NVSM_DT_TIMING
extern "C" int nvsm_debug_dt_ticks(unsigned long long* out, int n) {
    (void)hipDeviceSynchronize();
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(cunvsm::g_dt_ticks), sizeof(unsigned long long) * static_cast<size_t>(n)));
}
#endif
