// The two batch-sized projection products of a step on the bf16 matrix pipe, at fp32 accuracy:
//   forward   pre[B][d_e]     = phrase[B][d_w] · Tt[d_w][d_e]  (+ bias, + batch-norm column sums)   cpp/params.cu:417
//   backward  gphrase[B][d_w] = alpha · dx[B][d_e] · T          (+ per-row sums of squares,          cpp/objective.cu:453
//                                                                + the batch-norm backward on the way in: cpp/cudnn_utils.cu:143-183)
// gfx950 multiplies fp32 operands in its matrix cores at the fp32 VECTOR rate (v_mfma_f32_32x32x2_f32: 157 TFLOP/s, 1/16 of
// the bf16 rate) and has no TF32-like mode. But an fp32 number is EXACTLY the sum of three bf16 numbers — x = h + m + l with
// h = bf16(x), m = bf16(x − h), l = x − h − m (round to nearest; both differences are exact and the last has at most eight
// significant bits) — so a product a·b is
// exactly the sum of the nine products of their pieces, each of which the bf16 MFMA forms exactly (8 x 8 bits) and adds into
// an fp32 accumulator: fp32 accumulation of exact products, as the fp32 pipe does, in a different order. Six products
// (NVSM_GEMM_SPLIT=6, the default) leave out m·l, l·m and l·l, each below 2^-26 of a·b; NVSM_GEMM_SPLIT=9 keeps all nine;
// 0 switches this kernel off (exact-fp32 MFMA kernels: gemm_tstat / gemm_rows). Measured against an fp64 product (tools/exp/
// gemm_accuracy.py, M = 51 200, errors relative to Σ|a·b|; the table of DESIGN.md §4.1 is this run): backward shape, operands
// spread over 16 / 9 binades: exact-fp32 kernels max 1.35e-6 / rms 9.73e-8, nine products 1.04e-6 / 7.37e-8, six products
// 1.04e-6 / 7.37e-8; forward shape, N(0,1) x 0.1 N(0,1): 3.63e-7 / 2.80e-8, 3.28e-7 / 2.295e-8, 3.28e-7 / 2.295e-8 (six and
// nine agree to four digits) — the bf16 pipe's wider internal sum makes both forms slightly MORE accurate than the k-ordered
// fp32 chain. tests/test_gpu_parity.py::test_gemm_split_bf16_is_fp32_accurate asserts that relation up to M = 51 200.
//
// Shape of the work: M = batch is huge (51 200), N and K are a few hundred. One workgroup (eight waves, two per SIMD, 256
// registers each: the compiler keeps everything in arch VGPRs — a 512-register, one-wave-per-SIMD form of the same loop
// compiled to a v_accvgpr move per MFMA) per CU owns the CU's share of the rows — 51 200 / 256 = 200 rows = 12.5 blocks of 16 —
// and ALL columns; each wave owns a slice of the columns for all those rows (forward: 16 column blocks = 2 per wave; backward:
// 19 = 3, 3, 3, 2 | 2, 2, 2, 2, waves w and w + 4 share a SIMD: 5, 5, 5, 4 per SIMD; a wave runs the loop compiled for its
// own number of blocks), so the SIMDs carry the same number of MFMAs to within one column block. K runs in tiles of 32:
//   * the rows' tile of A is loaded from global memory once (fp32, 16 B per lane, two tiles ahead), cut into its three bf16
//     planes and stored to LDS in fragment order (row pitch 64 B, the octets of a row rotated so that the 16 B fragment reads
//     of every lane group of a ds_read_b128 hit all banks once: split_slot); two LDS images, one barrier per tile; the cutting rides between the MFMAs of the tile before;
//   * B — the projection matrix, 77 k elements, the same for every workgroup — arrives already cut: gemm_split_planes_kernel
//     writes its planes in fragment order (behind the projection update, off the critical path), a wave fetches the K tile of
//     ITS columns a tile ahead with three 16 B loads per block, L2-resident;
//   * operands are fed swapped (the tile is computed transposed), so a lane ends up with four consecutive columns of one
//     output row: 16 B stores, row sums of squares by two cross-lane adds, column sums by a 16-lane DPP reduce;
//   * column sums (batch-norm statistics) leave through the ordered grid-wide sum, row sums of squares are completed across
//     the waves through LDS in wave order: one value per row, no launch_sum_parts behind the product;
//   * the backward product applies the batch-norm backward to the rows of dy as it stages them and writes dx back for the
//     dT product (every element of A is staged by exactly one workgroup): launch_bn_dx + GEMM + launch_sum_parts in one launch.
// Alone, idle GPU, M = 51 200 (tools/exp/gemm_time.py): forward 59 us / backward 60 us with six products (74 / 74 with nine)
// against 90 / 95 us for the exact-fp32 kernels. The K loop runs within 7 % of what v_mfma_f32_16x16x32_bf16 delivers
// (19.3 cycles per instruction and SIMD; tools/exp/split_times.py); what is left is prologue (3 us: the first tile of A comes
// from HBM), epilogue (52 MB of stores at the end: 4 us) and the ordered column sums (4-10 us for the last workgroup).
#include "kernels.h"
#include "device_utils.h"

#include <atomic>
#include <cstdlib>

namespace cunvsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSplitPitch = 64;                                 // bytes per row of a plane: 32 bf16
// Octet q (eight consecutive k = 16 B) of row r sits at slot (q + 2 (r / 4)) mod 4 of the row's 64 B: ds_read_b128 is serviced
// in four groups of sixteen lanes that mix two octets — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md,
// LDS) — and with this rotation the sixteen 16 B words of every group fall on sixteen different bank quads (a padded pitch of
// 80 B, conflict-free for sixteen CONSECUTIVE lanes, cost every fragment read a second cycle per group: SQ_LDS_BANK_CONFLICT
// was half of SQ_LDS_IDX_ACTIVE).
__device__ __forceinline__ int split_slot(int r, int q) { return (q + 2 * ((r >> 2) & 3)) & 3; }
constexpr int kSplitWaves = 8;                                  // two per SIMD: 256 registers each, no AGPR shuffling
constexpr int kSplitMaxWaves = kSplitWaves;
// RBP = 16-row blocks per workgroup and pass: plane = 16 RBP rows, an image = three planes, two images
constexpr size_t split_lds_bytes(int rbp) {      // two images | column sums [2][np <= 320] | row sums of squares [waves][16 rbp] | batch-norm constants
    return static_cast<size_t>(2) * 3 * rbp * 16 * kSplitPitch + 2 * 320 * 4 + static_cast<size_t>(kSplitWaves) * rbp * 16 * 4 + 4 * 320 * 4;      // | PRE constants [4][K <= 320]
}
constexpr int kSplitMaxDevices = 64;

struct SplitArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    float alpha;
    const float* bias_n;
    double* colstats;        // [2][N] = Σ_rows C, Σ_rows C² or null
    GridSumWs sums;
    float* rowsq;            // [M] = rowsq_scale · Σ_cols C² per row (complete), or null
    float rowsq_scale;
    int nblocks;             // 16-row blocks = ceil(M / 16)
    int cb0[kSplitMaxWaves]; // first 16-column block of each wave
    int ncb[kSplitMaxWaves]; // its number of column blocks
    int np;                  // 16 · (column blocks in all)
    const unsigned char* planes;   // B cut into its three bf16 planes: [3][ceil(K / 32)][np][32] (gemm_split_planes_kernel)
    float* dump;
    int nt_store;            // the output with non-temporal stores
    // PRE: batch-norm backward on the rows of A (= dy, overwritten with dx) as they are staged (launch_bn_dx's job), or, with
    // pre == null, only the bias gradient grad_bias = (float) bn_sums (launch_colsum_finalize's)
    float* A_rw; const float* pre; const float* mean; const float* inv_std; const double* bn_sums;
    float* dbeta; float* dgamma; float* grad_bias; float inv_n;
};

#ifdef NVSM_SPLIT_TIMING
// experiments (make dbg; tools/exp/split_times.py): wall-clock stamps (100 MHz) of wave 0 of every workgroup of the last launch
__device__ unsigned long long g_split_times[1024 * 8];
#define SPLIT_STAMP(slot) do { if (threadIdx.x == 0) g_split_times[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
// shader-clock stamps inside the K loop of workgroup 0, waves 0 and 4 (one SIMD), first pass: [wave / 4][kt][point]
#define SPLIT_TICK(ps, kt, point) do { if (blockIdx.x == 0 && (ps) == 0 && (threadIdx.x & 255) == 0 && (kt) < 16) \
    g_split_times[4096 + (threadIdx.x >> 8) * 128 + (kt) * 8 + (point)] = clock64(); } while (0)
#else
#define SPLIT_STAMP(slot) do {} while (0)
#define SPLIT_TICK(ps, kt, point) do {} while (0)
#endif
constexpr int kSplitEpiStats = 1, kSplitEpiRowsq = 2, kSplitEpiBias = 4;

// a - b as ONE v_sub_f32: as vector arithmetic (or SLP-vectorised) the two subtractions of a pair become v_pk_add_f32, and
// packed fp32 VALU is slow next to MFMAs (MI355X_MICROARCH.md: "packed f32 VALU ... an anti-lever beside MFMAs"; gemm_dt.hip)
__device__ __forceinline__ float split_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x0, x1 -> one 32-bit word per plane, x0's piece in the lower half (element 2j of a fragment), x1's in the upper
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    // round-to-nearest pieces (v_cvt_pk_bf16_f32: two elements per instruction): h = bf16(x), m = bf16(x - h), l = x - h - m.
    // Both differences are exact in fp32 and the last one has at most eight significant bits, so x = h + m + l exactly;
    // the pieces are at most half an ulp of the piece above them (a cut by truncation leaves up to a whole one, all of x's
    // sign): the partial products a six-product run leaves out are below 2^-26 of a·b and of either sign.
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
    const float r0 = split_sub(x0, __uint_as_float(h << 16)), r1 = split_sub(x1, __uint_as_float(h & 0xffff0000u));
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
    const float s0 = split_sub(r0, __uint_as_float(m << 16)), s1 = split_sub(r1, __uint_as_float(m & 0xffff0000u));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}

struct SplitFrag { u32x4 h, m, l; };

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Σ over the 16 lanes of a DPP row (lanes with the same l >> 4); the total lands in lane 15 of the row
__device__ __forceinline__ float split_row16_sum(float v) {
    v += dpp_mov<0x111, 0xf>(v);
    v += dpp_mov<0x112, 0xf>(v);
    v += dpp_mov<0x114, 0xf>(v);
    v += dpp_mov<0x118, 0xf>(v);
    return v;
}

// The K loop is branch-free and issues the same loads in the same order every turn (past the last tile: harmless repeats of
// the last one), so that the compiler's wait-count bookkeeping comes out exact: vmcnt(NLD - 1 + NB) where a chunk of A is
// cut, vmcnt(NLD) where the planes of B are taken over. With the loads under `if (kt + 1 < KT)` it merged the branches'
// states conservatively and waited for the tile it had just requested (vmcnt(0) between the loads of A and B of every turn).
// (Loads as inline asm with hand-counted waits give the same code but are not safe: the register allocator may copy an asm
//  output before its data has arrived — a build with extra branches around the loads faulted.)
__device__ __forceinline__ void split_gload(u32x4& dst, unsigned voff, const void* base) {
    dst = *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(base) + voff);
}
template <int OFF>
__device__ __forceinline__ void split_gload_off(u32x4& dst, unsigned voff, const void* base) {
    dst = *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(base) + voff + OFF);
}

// The work of one wave: CBW column blocks (all of them multiplied), RBP row blocks per pass (the accumulators: 4 RBP CBW
// registers); NPROD 9 or 6.
template <int CBW, int RBP, int NPROD, int EPI, bool PRE>
__device__ __forceinline__ void split_body(const SplitArgs& g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char split_lds[];
    constexpr int WAVES = kSplitWaves;
    constexpr int T = WAVES * 64;
    constexpr int kRows = RBP * 16, kPlane = kRows * kSplitPitch, kStage = 3 * kPlane;
    constexpr int kF4 = kRows * 8;                            // float4s of a tile of A
    constexpr int NLD = (kF4 + T - 1) / T;                    // ... per thread
    constexpr int STEP = (RBP - 1) / NLD > 0 ? (RBP - 1) / NLD : 1;      // staging chunk u rides with row block 1 + u STEP
    static_assert(1 + (NLD - 1) * STEP < RBP, "staging chunks must fit the row-block loop");
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgs = gridDim.x, me = blockIdx.x;
    const int base = g.nblocks / wgs, extra = g.nblocks % wgs;
    const int rb_begin = me * base + (me < extra ? me : extra);
    const int nb = base + (me < extra ? 1 : 0);
    const int cb0 = g.cb0[w], ncb = g.ncb[w];
    const int KT = (g.K + 31) / 32;
    const size_t plane_stride = static_cast<size_t>(KT) * g.np * 64;
    const unsigned b_voff = static_cast<unsigned>((16 * cb0 + i) * 64 + q * 16);      // this lane inside a K tile of a plane
    const unsigned a_last = static_cast<unsigned>((static_cast<size_t>(g.M - 1) * g.lda + g.K) * 4 - 16);      // last float4 of A

    SPLIT_STAMP(0);
    // LDS: two images of a tile of A (three planes each) | column sums [2][np <= 320] | row sums of squares [WAVES][kRows]
    float* st = reinterpret_cast<float*>(split_lds + 2 * kStage);
    float* rs_lds = st + 2 * 320;
    float* consts = rs_lds + WAVES * kRows;                   // PRE: [4][K <= 320] μ, invσ, dβ, dγ
    int* sum_flag = reinterpret_cast<int*>(split_lds);
    if (EPI & kSplitEpiStats) {
        for (int k = tid; k < 2 * g.np; k += T) st[k] = 0.f;       // (the barriers of the K loop order this before the first add)
    }

    if (PRE) {
        for (int k = tid; k < g.K; k += T) {
            consts[k] = g.mean[k]; consts[g.K + k] = g.inv_std[k];
            const float db = static_cast<float>(g.bn_sums[k]), dg = static_cast<float>(g.bn_sums[g.K + k]);      // cudnn_utils.cu:158-173
            consts[2 * g.K + k] = db; consts[3 * g.K + k] = dg;
            if (blockIdx.x == 0) { g.dbeta[k] = db; g.dgamma[k] = dg; g.grad_bias[k] = db; }      // ∂β is the bias gradient; ∂γ is dropped (:173)
        }
        __syncthreads();
    }
    if (!PRE && g.grad_bias && blockIdx.x == 0)      // no batch-norm: the bias gradient is Σdy
        for (int k = tid; k < g.K; k += T) g.grad_bias[k] = static_cast<float>(g.bn_sums[k]);

    // passes of at most RBP row blocks, as even as they go (13 blocks at RBP = 7: 7 + 6)
    const int npass = (nb + RBP - 1) / RBP;
    for (int ps = 0, b0 = 0; ps < npass; ++ps) {
        const int nbb = nb / npass + (ps < nb % npass ? 1 : 0);
        const int row0 = (rb_begin + b0) * 16;
        b0 += nbb;
        const int nrows = (g.M - row0) < nbb * 16 ? (g.M - row0) : nbb * 16;       // >= 1

        // byte offsets of this thread's float4s inside a K tile of A (rows past the end: the last row, never stored)
        unsigned a_voff[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + T * u;
            int row = idx >> 3;
            row = row < nrows ? row : nrows - 1;
            a_voff[u] = static_cast<unsigned>((static_cast<size_t>(row0 + row) * g.lda + 4 * (idx & 7)) * 4);
        }
        // (tiles past the last one: the last one again; a tile may reach past K: those float4s are zeroed in store_a and the
        //  address stays inside the matrix)
        auto load_a1 = [&](int kt, int u, u32x4& r, u32x4& x) {
            const unsigned o = a_voff[u] + 128u * static_cast<unsigned>(kt < KT ? kt : KT - 1);
            split_gload(r, o < a_last ? o : a_last, g.A);
            if (PRE) split_gload(x, o < a_last ? o : a_last, g.pre);
        };
        auto store_a1 = [&](int kt, int stage, int u, const u32x4& r, const u32x4& x) {
            const int idx = tid + T * u;
            if (idx < kF4) {
                const int row = idx >> 3, k4 = idx & 7;
                const bool ok = (row < nrows) && (32 * kt + 4 * k4 < g.K);
                u32x4 v = ok ? r : u32x4{0u, 0u, 0u, 0u};
                if (PRE) {
                    // dx = invσ · (dy − (dβ + x̂·dγ) / N), x̂ = (x − μ)·invσ      (bn_dx_kernel, loss_bn.hip); written back over dy
                    const int k = ok ? 32 * kt + 4 * k4 : 0;
                    const float4 mu = *reinterpret_cast<const float4*>(consts + k), is = *reinterpret_cast<const float4*>(consts + g.K + k);
                    const float4 db = *reinterpret_cast<const float4*>(consts + 2 * g.K + k), dg = *reinterpret_cast<const float4*>(consts + 3 * g.K + k);
                    const float mu_[4] = {mu.x, mu.y, mu.z, mu.w}, is_[4] = {is.x, is.y, is.z, is.w};
                    const float db_[4] = {db.x, db.y, db.z, db.w}, dg_[4] = {dg.x, dg.y, dg.z, dg.w};
                    f32x4 d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xhat = (__uint_as_float(x[e]) - mu_[e]) * is_[e];
                        d[e] = is_[e] * (__uint_as_float(v[e]) - (db_[e] + xhat * dg_[e]) * g.inv_n);
                    }
                    *reinterpret_cast<f32x4*>(ok ? g.A_rw + (static_cast<size_t>(row0 + row) * g.lda + 32 * kt + 4 * k4) : g.dump) = d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? __float_as_uint(d[e]) : 0u;
                }
                unsigned h0, m0, l0, h1, m1, l1;
                split_pair(__uint_as_float(v[0]), __uint_as_float(v[1]), h0, m0, l0);
                split_pair(__uint_as_float(v[2]), __uint_as_float(v[3]), h1, m1, l1);
                unsigned char* p = split_lds + stage * kStage + row * kSplitPitch + split_slot(row, k4 >> 1) * 16 + (k4 & 1) * 8;
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + kPlane) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(p + 2 * kPlane) = make_uint2(l0, l1);
            }
        };
        // the wave's column blocks of K tile kt: three 16 B loads per block from the planes of B (gemm_split_planes_kernel;
        // zeros outside the matrix), a KB per 16 lanes
        auto load_bf = [&](int kt, SplitFrag (&f)[CBW]) {
            const unsigned char* ph = g.planes + (static_cast<size_t>(kt < KT ? kt : KT - 1) * g.np) * 64;      // wave-uniform: SGPR bases
            const unsigned char* pm = ph + plane_stride;
            const unsigned char* pl = pm + plane_stride;
            split_gload(f[0].h, b_voff, ph); split_gload(f[0].m, b_voff, pm); split_gload(f[0].l, b_voff, pl);
            if constexpr (CBW > 1) { split_gload_off<1024>(f[1].h, b_voff, ph); split_gload_off<1024>(f[1].m, b_voff, pm); split_gload_off<1024>(f[1].l, b_voff, pl); }
            if constexpr (CBW > 2) { split_gload_off<2048>(f[2].h, b_voff, ph); split_gload_off<2048>(f[2].m, b_voff, pm); split_gload_off<2048>(f[2].l, b_voff, pl); }
            static_assert(CBW <= 3, "column blocks per wave");
        };

        f32x4 acc[RBP][CBW];
#pragma unroll
        for (int rb = 0; rb < RBP; ++rb)
#pragma unroll
            for (int c = 0; c < CBW; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- prologue: tile 0 of A into image 0 and of B into registers; tile 1 of A in flight ----
        u32x4 ar[NLD], xr[NLD];
        SplitFrag bf[CBW], bfn[CBW];
#pragma unroll
        for (int u = 0; u < NLD; ++u) load_a1(0, u, ar[u], xr[u]);
        load_bf(0, bf);
#pragma unroll
        for (int u = 0; u < NLD; ++u) store_a1(0, 0, u, ar[u], xr[u]);
#pragma unroll
        for (int u = 0; u < NLD; ++u) load_a1(1, u, ar[u], xr[u]);
        __syncthreads();
        if (ps == 0) SPLIT_STAMP(1);

        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            // in flight here: the NLD chunks of tile kt + 1 of A (requested during the previous turn). This wave's planes of B
            // for the next tile go out now and are waited for at the bottom of the turn.
            SPLIT_TICK(ps, kt, 0);
            load_bf(kt + 1, bfn);
            const unsigned char* sb = split_lds + cur * kStage + i * kSplitPitch + split_slot(i, q) * 16;      // (row blocks start at multiples of 16)
            // the fragments of row block rb + 1 are read from LDS while block rb is multiplied (the scheduling barriers pin
            // that order: left alone the compiler hoists all the blocks' reads to the top and spills)
            u32x4 ah = *reinterpret_cast<const u32x4*>(sb);
            u32x4 am = *reinterpret_cast<const u32x4*>(sb + kPlane);
            u32x4 al = *reinterpret_cast<const u32x4*>(sb + 2 * kPlane);
#pragma unroll
            for (int rb = 0; rb < RBP; ++rb) {
#ifndef NVSM_SPLIT_NO_PRIO
                // Round 6: the wave that is FURTHER along in the tile yields — s_setprio falls with the row block, in four levels —
                // so that the two waves of a SIMD stay within a few row blocks of each other. The matrix pipe serves the older wave
                // first: left alone, wave w had issued its 156 MFMAs of a tile when wave w + 4 had issued half of its own, waited
                // 1 850 cycles at the tile's barrier while its partner finished ALONE at three quarters of the pipe's rate (its fragment
                // reads and staging exposed), 7 000 cycles per tile; with the priorities 6 500 (tools/exp/split_times.py,
                // profiles/r06_exp_split_times*.txt): forward product 57.2 -> 56.3 us alone, the 51 200-window step 0.8604 -> 0.8577 ms.
                {
                    const int level = rb * 4 / RBP, before = rb > 0 ? (rb - 1) * 4 / RBP : -1;
                    if (level != before) {
                        if (level == 0) __builtin_amdgcn_s_setprio(3);
                        else if (level == 1) __builtin_amdgcn_s_setprio(2);
                        else if (level == 2) __builtin_amdgcn_s_setprio(1);
                        else __builtin_amdgcn_s_setprio(0);
                    }
                }
#endif
                u32x4 nh = ah, nm = am, nl = al;
                if (rb + 1 < RBP) {
                    const unsigned char* ap = sb + (rb + 1) * 16 * kSplitPitch;
                    nh = *reinterpret_cast<const u32x4*>(ap);
                    nm = *reinterpret_cast<const u32x4*>(ap + kPlane);
                    nl = *reinterpret_cast<const u32x4*>(ap + 2 * kPlane);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef NVSM_SPLIT_LATE_BARRIER
                // Round 6: the tile's ONE barrier stands HERE, in front of the last row block's MFMAs, not at the end of the tile. What
                // it has to separate is already separate at this point — this wave's staging stores of tile kt + 1 (the last rides with
                // row block 1 + (NLD - 1) STEP <= RBP - 2) and its fragment reads of tile kt's image (the last, for this row block,
                // were issued one row block ago; lgkmcnt(0) below) — so behind it image cur may be overwritten and image cur ^ 1 read.
                // At the end of the tile every wave then runs straight on into the next one (planes of B taken over, first fragment
                // reads) in the shadow of this row block's twelve MFMAs instead of all eight waves meeting with an empty matrix pipe:
                // 6 500 -> 6 1xx cycles per tile (tools/exp/split_times.py).
                if (rb == RBP - 1) {
                    static_assert(1 + (NLD - 1) * STEP <= RBP - 2, "the last staging chunk must be stored before the barrier's row block");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
                // staging of the next tile rides with the MFMAs: chunk u of tile kt + 1 is cut into the other image (its
                // readers passed the barrier that ended the previous turn) and chunk u of tile kt + 2 requested. Behind
                // the chunk's load sit the later chunks of its tile, the planes of B and the earlier chunks of tile kt + 2.
                if ((rb - 1) % STEP == 0 && (rb - 1) / STEP < NLD && rb >= 1) {
                    const int u = (rb - 1) / STEP;
                    store_a1(kt + 1, cur ^ 1, u, ar[u], xr[u]);
                    load_a1(kt + 2, u, ar[u], xr[u]);
                }
                if (rb < nbb) {
                    // smallest products first; consecutive MFMAs go to different accumulators. (Column blocks the wave does
                    // not own are multiplied as zeros: the widest slice sets the pace anyway.)
                    if (NPROD == 9) {
#pragma unroll
                        for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].l, al, acc[rb][c]);
#pragma unroll
                        for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].l, am, acc[rb][c]);
#pragma unroll
                        for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].m, al, acc[rb][c]);
                    }
#pragma unroll
                    for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].l, ah, acc[rb][c]);
#pragma unroll
                    for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].h, al, acc[rb][c]);
#pragma unroll
                    for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].m, am, acc[rb][c]);
#pragma unroll
                    for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].m, ah, acc[rb][c]);
#pragma unroll
                    for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].h, am, acc[rb][c]);
#pragma unroll
                    for (int c = 0; c < CBW; ++c) acc[rb][c] = mfma_bf16(bf[c].h, ah, acc[rb][c]);
                }
                if (rb == 0) SPLIT_TICK(ps, kt, 4);
                if (rb == RBP / 2) SPLIT_TICK(ps, kt, 5);
                __builtin_amdgcn_sched_barrier(0);
                ah = nh; am = nm; al = nl;
            }
            SPLIT_TICK(ps, kt, 1);
            // (behind the planes: the NLD chunks of tile kt + 2)
#pragma unroll
            for (int c = 0; c < CBW; ++c) bf[c] = bfn[c];
            SPLIT_TICK(ps, kt, 2);
#ifdef NVSM_SPLIT_LATE_BARRIER
            __syncthreads();
#endif
            SPLIT_TICK(ps, kt, 3);
        }
        if (ps == 0) SPLIT_STAMP(2);
        if (ps == npass - 1) SPLIT_STAMP(3);
        // ---- epilogue: acc[rb][c][r] = C[row0 + 16 rb + i][16 (cb0 + c) + 4 q + r] ----
        float cs1[CBW][4], cs2[CBW][4];                             // column sums of this lane's values of this pass
#pragma unroll
        for (int c = 0; c < CBW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) { cs1[c][r] = 0.f; cs2[c][r] = 0.f; }
#pragma unroll
        for (int rb = 0; rb < RBP; ++rb) {
            if (rb < nbb) {
                const int row = row0 + 16 * rb + i;
                const bool row_ok = row < g.M;
                float rsq = 0.f;
#pragma unroll
                for (int c = 0; c < CBW; ++c) {
                    if (c < ncb) {
                        const int col = 16 * (cb0 + c) + 4 * q;
                        const bool ok = row_ok && col < g.N;                 // N % 4 == 0
                        float v[4];
                        if (EPI & kSplitEpiBias) {
                            const float4 bb = *reinterpret_cast<const float4*>(g.bias_n + (col < g.N ? col : 0));
                            v[0] = g.alpha * acc[rb][c][0] + bb.x; v[1] = g.alpha * acc[rb][c][1] + bb.y;
                            v[2] = g.alpha * acc[rb][c][2] + bb.z; v[3] = g.alpha * acc[rb][c][3] + bb.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = g.alpha * acc[rb][c][r];
                        }
                        f32x4* dstp = reinterpret_cast<f32x4*>(ok ? g.C + static_cast<size_t>(row) * g.ldc + col : g.dump);
                        const f32x4 vv = f32x4{v[0], v[1], v[2], v[3]};
                        if (g.nt_store) __builtin_nontemporal_store(vv, dstp); else *dstp = vv;
                        if (EPI & kSplitEpiRowsq) rsq += ok ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
                        if (EPI & kSplitEpiStats) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float x = ok ? v[r] : 0.f;
                                cs1[c][r] += x;
                                cs2[c][r] += x * x;
                            }
                        }
                    }
                }
                if (EPI & kSplitEpiRowsq) {
                    rsq += __shfl_xor(rsq, 16);
                    rsq += __shfl_xor(rsq, 32);
                    if (q == 0) rs_lds[w * kRows + 16 * rb + i] = rsq;
                }
            }
        }
        if (EPI & kSplitEpiStats) {
            // the lane's sums over its rows -> sums over the 16 rows of a lane group (lane 15) -> the column's LDS word
#pragma unroll
            for (int c = 0; c < CBW; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s1 = split_row16_sum(cs1[c][r]);
                    const float s2 = split_row16_sum(cs2[c][r]);
                    if (i == 15 && c < ncb) {
                        st[16 * (cb0 + c) + 4 * q + r] += s1;
                        st[g.np + 16 * (cb0 + c) + 4 * q + r] += s2;
                    }
                }
            }
        }
        if (EPI & kSplitEpiRowsq) {
            __syncthreads();
            for (int t = tid; t < nbb * 16; t += T) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) s += rs_lds[ww * kRows + t];
                if (row0 + t < g.M) g.rowsq[row0 + t] = s * g.rowsq_scale;
            }
            __syncthreads();
        }
    }

    SPLIT_STAMP(4);
    if (EPI & kSplitEpiStats) {
        const int np = g.np;
        __syncthreads();
        const GridSumWs& ws = g.sums;
        auto val = [&](int k) -> float { return st[k]; };
        double* cs = g.colstats;
        const int N = g.N;
        auto out = [&](int k, double v) {
            const int s = k / np, n = k - s * np;
            if (n < N) cs[static_cast<size_t>(s) * N + n] = v;
        };
        grid_sum_ordered<T>(ws.part, ws.part2, ws.arrive, ws.fan, 2 * np, me, wgs, val, out, sum_flag);
    }
    SPLIT_STAMP(5);
}

// Eight waves; a wave owns CBW or (MIXED) CBW - 1 column blocks and runs the body compiled for that number — the same K loop
// and the same barriers either way.
template <int CBW, int RBP, int NPROD, int EPI, bool MIXED, bool PRE>
__global__ __launch_bounds__(kSplitWaves * 64) void gemm_split_kernel(SplitArgs g) {
    if constexpr (MIXED) {
        const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
        if (g.ncb[w] == CBW) split_body<CBW, RBP, NPROD, EPI, PRE>(g);
        else split_body<CBW - 1, RBP, NPROD, EPI, PRE>(g);
    } else {
        split_body<CBW, RBP, NPROD, EPI, PRE>(g);
    }
}

template <int CBW, int RBP, int NPROD, int EPI, bool MIXED, bool PRE = false>
static bool split_launch_epi(const SplitArgs& g, int wgs, hipStream_t s) {
    constexpr size_t kSplitLdsBytes = split_lds_bytes(RBP);
    static std::atomic<bool> attr_set[kSplitMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSplitMaxDevices) return false;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<CBW, RBP, NPROD, EPI, MIXED, PRE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSplitLdsBytes)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev].store(true, std::memory_order_release);
    }
    (void)hipGetLastError();
    NVSM_LAUNCH((gemm_split_kernel<CBW, RBP, NPROD, EPI, MIXED, PRE>), dim3(wgs), dim3(kSplitWaves * 64), kSplitLdsBytes, s, g);
    return hipGetLastError() == hipSuccess;
}

// the forward product carries batch-norm column sums or a bias, the backward one row sums of squares or nothing
template <int CBW, int RBP, int NPROD, bool FORWARD>
static bool split_launch(const SplitArgs& g, int wgs, hipStream_t s) {
    const int epi = (g.colstats ? kSplitEpiStats : 0) | (g.rowsq ? kSplitEpiRowsq : 0) | (g.bias_n ? kSplitEpiBias : 0);
    if constexpr (FORWARD) {
        if (epi == kSplitEpiStats) return split_launch_epi<CBW, RBP, NPROD, kSplitEpiStats, false>(g, wgs, s);
        if (epi == kSplitEpiBias) return split_launch_epi<CBW, RBP, NPROD, kSplitEpiBias, false>(g, wgs, s);
        if (epi == 0) return split_launch_epi<CBW, RBP, NPROD, 0, false>(g, wgs, s);
    } else {
        if (g.pre) {
            if (epi == kSplitEpiRowsq) return split_launch_epi<CBW, RBP, NPROD, kSplitEpiRowsq, true, true>(g, wgs, s);
            if (epi == 0) return split_launch_epi<CBW, RBP, NPROD, 0, true, true>(g, wgs, s);
        }
        if (epi == kSplitEpiRowsq) return split_launch_epi<CBW, RBP, NPROD, kSplitEpiRowsq, true>(g, wgs, s);
        if (epi == 0) return split_launch_epi<CBW, RBP, NPROD, 0, true>(g, wgs, s);
    }
    return false;
}

// B (the projection matrix, 77 k elements) cut into planes[3][ceil(K / 32)][np][32] bf16, zero outside the matrix: a thread per
// four consecutive k of one column. BLAY 0: B is [K][N] (ldb), 1: B is stored [N][K] (ldb).
template <int BLAY>
__global__ __launch_bounds__(256) void gemm_split_planes_kernel(const float* __restrict__ B, int N, int K, int ldb, int np,
                                                                unsigned char* __restrict__ planes) {
    const int KT = (K + 31) / 32;
    const int total = KT * np * 8;
    const size_t plane_stride = static_cast<size_t>(KT) * np * 64;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        // BLAY 0: consecutive threads walk n (coalesced rows of B); BLAY 1: consecutive threads walk k (rows of the stored matrix)
        int kt, n, k4;
        if (BLAY == 0) { n = idx % np; const int r = idx / np; k4 = r & 7; kt = r >> 3; }
        else { k4 = idx & 7; const int r = idx >> 3; kt = r % KT; n = r / KT; }
        const int k0 = 32 * kt + 4 * k4;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = n < N && k0 + e < K;
            const size_t off = ok ? (BLAY == 0 ? static_cast<size_t>(k0 + e) * ldb + n : static_cast<size_t>(n) * ldb + k0 + e) : 0;
            const float v = B[off];
            x[e] = ok ? v : 0.f;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        split_pair(x[0], x[1], h0, m0, l0);
        split_pair(x[2], x[3], h1, m1, l1);
        unsigned char* p = planes + (static_cast<size_t>(kt) * np + n) * 64 + k4 * 8;
        *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(p + 2 * plane_stride) = make_uint2(l0, l1);
    }
}

size_t gemm_split_planes_bytes(int N, int K) {
    // + 2 KB: a wave with fewer column blocks than its kernel's maximum still loads the maximum (the products are thrown
    // away), and behind the last block of the last tile of the last plane that reaches past the planes
    return static_cast<size_t>(3) * ((K + 31) / 32) * (16 * ((N + 15) / 16)) * 64 + 2048;
}
PlaneTarget gemm_split_plane_target(int N, int K, void* planes, int transposed) {
    const int np = 16 * ((N + 15) / 16), KT = (K + 31) / 32;
    return PlaneTarget{static_cast<unsigned char*>(planes), static_cast<size_t>(KT) * np * 64, 1, np, transposed};
}
void launch_gemm_split_planes(int b_layout, const float* B, int N, int K, int ldb, void* planes, hipStream_t s) {
    const int np = 16 * ((N + 15) / 16), total = ((K + 31) / 32) * np * 8;
    const int grid = (total + 255) / 256;
    if (b_layout == 0) NVSM_LAUNCH((gemm_split_planes_kernel<0>), dim3(grid), dim3(256), 0, s, B, N, K, ldb, np, static_cast<unsigned char*>(planes));
    else NVSM_LAUNCH((gemm_split_planes_kernel<1>), dim3(grid), dim3(256), 0, s, B, N, K, ldb, np, static_cast<unsigned char*>(planes));
}

// NVSM_GEMM_SPLIT: 0 = never, 6 (default) / 9 = number of partial products (tuning.h: per handle)
int gemm_split_products() {
    return tuning().gemm_split;
}

// what launch_gemm_split accepts (given 16 B aligned operands and leading dimensions that are multiples of 4)
bool gemm_split_covers(int b_layout, int M, int N, int K, bool bn) {
    if (!gemm_split_products() || M < 1024 || (K % 4) || (N % 4) || K < 8) return false;
    if (static_cast<uint64_t>(M) * static_cast<uint64_t>(K) * 4 >= (1ull << 32)) return false;      // 32-bit byte offsets into A (lda = K)
    const int cbs = (N + 15) / 16;
    if (bn && (b_layout != 1 || K > 320)) return false;
    return b_layout == 0 ? cbs == 2 * kSplitWaves : (cbs > 2 * kSplitWaves && cbs <= 3 * kSplitWaves);
}

// returns false when the shape is not one this kernel covers (nothing launched). rowsq: ONE complete value per row.
// ws: the planes of B (cut here, on `s`, unless ws->ready says they are current).
bool launch_gemm_split(int b_layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       float alpha, const float* bias_n, hipStream_t s, double* colstats, const GridSumWs* sums, float* rowsq,
                       float rowsq_scale, GemmSplitWs* ws, const BnDxFused* bn) {
    const int nprod = gemm_split_products();
    if (bn && (b_layout != 1 || K > 320 || (bn->pre && (bn->dy != A || reinterpret_cast<uintptr_t>(bn->pre) % 16)))) return false;
    if (!nprod || M < 1024 || !ws || !ws->planes || ws->bytes < gemm_split_planes_bytes(N, K)) return false;
    // (the kernel addresses A — and pre — by 32-bit byte offsets from the base)
    if (static_cast<uint64_t>(M) * static_cast<uint64_t>(lda) * 4 >= (1ull << 32) || lda < K) return false;
    if ((K % 4) || (N % 4) || (lda % 4) || (ldb % 4) || (ldc % 4) || K < 8) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) % 16) return false;
    if (bias_n && reinterpret_cast<uintptr_t>(bias_n) % 16) return false;
    const int cbs = (N + 15) / 16;
    static std::atomic<int> cus_of[kSplitMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSplitMaxDevices) return false;
    int num_cus = cus_of[dev].load(std::memory_order_acquire);
    if (num_cus == 0) {
        hipDeviceProp_t prop;
        num_cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus_of[dev].store(num_cus, std::memory_order_release);
    }
    float* dump = gemm_dump_buffer();
    if (!dump) return false;
    SplitArgs g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.bias_n = bias_n; g.colstats = colstats; g.rowsq = rowsq; g.rowsq_scale = rowsq_scale;
    if (bn) {
        g.A_rw = bn->dy; g.pre = bn->pre; g.mean = bn->mean; g.inv_std = bn->inv_std; g.bn_sums = bn->sums;
        g.dbeta = bn->dbeta; g.dgamma = bn->dgamma; g.grad_bias = bn->grad_bias; g.inv_n = static_cast<float>(1.0 / bn->n_global);
    }
    g.nt_store = tuning().split_nt;
    g.nblocks = (M + 15) / 16; g.np = 16 * cbs; g.dump = dump; g.planes = static_cast<const unsigned char*>(ws->planes);
    int wgs = num_cus < g.nblocks ? num_cus : g.nblocks;
    if (colstats) {
        if (!sums || sums->colgroups < 1 || sums->contrib_cap < wgs || sums->width_cap < 2 * g.np) return false;
        g.sums = *sums;
        g.sums.fan = grid_sum_fan(wgs);
        if ((wgs + g.sums.fan - 1) / g.sums.fan > sums->groups_cap) return false;
    }
    // column blocks to waves: as even as they go. Waves w and w + 4 of a workgroup share a SIMD (NVSM_SPLIT_DEAL=1 assumes
    // 2 j and 2 j + 1 do): the wider slices go to waves of different SIMDs first — 19 blocks = 3, 3, 3, 2 | 2, 2, 2, 2 is
    // 5, 5, 5, 4 per SIMD.
    auto deal = [&](int waves) {
        const bool adjacent = tuning().split_deal == 1;
        const int lo = cbs / waves, wide = cbs - lo * waves;
        int n_of[kSplitMaxWaves] = {0};
        for (int k = 0; k < waves; ++k) n_of[adjacent ? ((2 * k) % waves + (2 * k) / waves) : k] = lo + (k < wide ? 1 : 0);
        int c0 = 0;
        for (int w = 0; w < kSplitMaxWaves; ++w) { g.cb0[w] = c0; g.ncb[w] = n_of[w]; c0 += n_of[w]; }
        return lo + (wide ? 1 : 0);
    };
    bool ok = false;
    const int cbw = deal(kSplitWaves);
    const bool forward = b_layout == 0;
    if ((forward && (cbw != 2 || cbs % kSplitWaves)) || (!forward && (cbw != 3 || cbs < 2 * kSplitWaves))) return false;
    if (!ws->ready) { launch_gemm_split_planes(b_layout, B, N, K, ldb, ws->planes, s); ws->ready = true; }
    if (forward) {
        // the forward product: 16 column blocks = two per wave, thirteen row blocks (the whole share of a CU) in one pass
        ok = nprod == 9 ? split_launch<2, 13, 9, true>(g, wgs, s) : split_launch<2, 13, 6, true>(g, wgs, s);
    } else {
        // the backward one: 19 column blocks = 3, 3, 3, 2 | 2, 2, 2, 2 (waves w and w + 4 share a SIMD: 5, 5, 5, 4 per SIMD);
        // seven row blocks per pass keep a three-block wave's accumulators at 84 registers
        ok = nprod == 9 ? split_launch<3, 7, 9, false>(g, wgs, s) : split_launch<3, 7, 6, false>(g, wgs, s);
    }
    return ok;
}

}  // namespace cunvsm

#ifdef NVSM_SPLIT_TIMING
extern "C" int nvsm_debug_split_times(unsigned long long* out, int n) {
    (void)hipDeviceSynchronize();
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(cunvsm::g_split_times), sizeof(unsigned long long) * static_cast<size_t>(n)));
}
#endif
