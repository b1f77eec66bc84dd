// dT[d_w][d_e] = phraseᵀ[d_w x B] · dx[B x d_e], the projection gradient (cpp/params.cu:526-531), on the bf16 matrix pipe at
// fp32 accuracy (the split-bf16 arithmetic of gemm_split.hip: x = h + m + l exactly, six or nine bf16 MFMAs per product, fp32
// accumulation), split-K over the batch. Both operands are batch-sized and enter the product with the batch as the reduction
// dimension, i.e. transposed against the way they lie in memory. Round 3's kernel cut AND transposed every element in registers
// with all eight waves in step: ≈ 900 staging instructions per SIMD and K tile against 120 MFMAs, MfmaUtil 17 %, 114 us at
// the 51 200-window batch. This one:
//   * a workgroup (eight waves) owns a slab of the batch and one HALF of the d_e columns: 320 x 128 of the output. The two
//     workgroups of a slab run on the same XCD at the same time (block index → (slab, half) below), so the phrase rows, which
//     both need, come from HBM once and from that XCD's L2 the second time (measured: TCC hit rate 50 %, fabric reads = the
//     operands once);
//   * K runs in tiles of 16 batch rows. A thread fetches up to four float4s of the tile (fp32, coalesced along the rows, TWO
//     tiles ahead, two register sets), cuts each into its three bf16 pieces (v_cvt_pk_bf16_f32: 11 VALU per pair) and stores
//     them ROW-major into the tile's LDS image: per plane and 64-column group a 1 KB chunk = 8 rows x 128 B, the 16 B pieces
//     of a chunk permuted (piece c of row k at slot 8 k + (c ^ 4 (k >> 1 & 1)));
//   * the MFMA fragments, whose K runs along the batch, come out of that row-major image with ds_read_b64_tr_b16 — the 4 x 4
//     transposition happens in the LDS read path, not in registers — and with the permutation above the 4 rows x 64 B a
//     half-wave touches fall on all 64 banks once (SQ_LDS_BANK_CONFLICT = 0);
//   * wave (wm, wn) of a 2 x 4 grid owns 160 x 32 of the output: 5 blocks of v_mfma_f32_32x32x16_bf16, 80 accumulator
//     registers, the products PRODUCT-major so that consecutive MFMAs go to different accumulators;
//   * the two waves of a SIMD run in opposite phases (below): one multiplies while the other stages and reads; one barrier
//     per tile, two LDS images;
//   * the slab's partial product goes to partial[slab] and launch_splitk_reduce adds the slabs in order (deterministic).
// An intermediate form of this kernel took its operands as bf16 PLANES written by the projection products on the way (LDS-DMA
// straight into the image, no cutting here): 45.7 us alone — and 241 MB more HBM traffic per step (177 MB of planes written
// + 64 MB more read than fp32), which an HBM-bound step pays for in full: 0.92 -> 0.965 ms (git a07ffbc). fp32 operands cost
// the VALU of the cut and nothing else.
#include "../../include/cunvsm_amd.h"
#include "kernels.h"
#include "device_utils.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace cunvsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kDtWaves = 8, kDtThreads = kDtWaves * 64;
constexpr int kDtMaxGA = 5, kDtMaxGB = 4;              // 64-column groups of the two operands: M <= 320, N <= 256
constexpr int kDtHalfGB = 2;                           // ... of dx per workgroup: 128 columns
constexpr int kDtTileRows = 16;                        // batch rows per K tile = one MFMA k step
constexpr int kDtStage = 4;                            // float4s a thread stages per tile: ceil(16 x (80 + 32) / 512)
constexpr int kDtRB = 5;                               // 32 x 32 blocks per wave: 5 along the rows of the output (phrase columns) x 1
constexpr int kDtMaxDevices = 64;

struct DtArgs {
    const float* A; const float* B;                    // phrase [rows][M] (lda) / dx [rows][N] (ldb)
    int lda, ldb;
    float* P; size_t p_stride; int ldc;                // partial [slabs][M][N]
    int rows, M, N, slab_rows, slabs;                  // rows = batch (the reduction); slab_rows a multiple of 16
    int ga, gb;                                        // 64-column groups in use
    int halves;                                        // 1: N <= 128
};

#ifdef NVSM_DT_TIMING
// experiments (make dbg; tools/exp/dt_ticks.py): shader-clock stamps of waves 0, 4 (one SIMD), 1 and 7 of two workgroups inside the K loop
__device__ unsigned long long g_dt_ticks[2 * 4 * 16 * 8];
#define DT_TICK(t, point) do { const int _bw = blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 1 : -1); \
    const int _ww = w == 0 ? 0 : (w == 4 ? 1 : (w == 1 ? 2 : (w == 7 ? 3 : -1))); \
    if (_bw >= 0 && _ww >= 0 && lane == 0 && (t) < 16) g_dt_ticks[((_bw * 4 + _ww) * 16 + (t)) * 8 + (point)] = clock64(); } while (0)
#else
#define DT_TICK(t, point) do {} while (0)
#endif

// f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>): a loop whose index is a constant expression in the body
template <int B, int E, class F>
__device__ __forceinline__ void dt_static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); dt_static_for<B + 1, E>(f); }
}

// a - b as ONE v_sub_f32. (Written as vector arithmetic — or left to the SLP vectoriser — the two subtractions of a pair become
// v_pk_add_f32, and packed fp32 VALU runs on the matrix pipe's datapath: next to a partner wave's MFMAs the staging of a tile
// took 1 600 - 3 300 cycles instead of 770, MI355X_MICROARCH.md "packed f32 VALU ... an anti-lever beside MFMAs".)
__device__ __forceinline__ float dt_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x0, x1 -> one 32-bit word per plane, x0's piece in the lower half: round-to-nearest pieces (v_cvt_pk_bf16_f32: two elements per
// instruction) h = bf16(x), m = bf16(x - h), l = x - h - m; both differences are exact in fp32 and the last one has at most
// eight significant bits, so x = h + m + l exactly (gemm_split.hip split_pair)
__device__ __forceinline__ void dt_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
    const float r0 = dt_sub(x0, __uint_as_float(h << 16)), r1 = dt_sub(x1, __uint_as_float(h & 0xffff0000u));
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
    const float s0 = dt_sub(r0, __uint_as_float(m << 16)), s1 = dt_sub(r1, __uint_as_float(m & 0xffff0000u));
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t));
}

// FULL: every wave has all its blocks (ten row blocks, eight column blocks: 289 <= M <= 320, 225 <= N <= 256 — the projection's
// shape): no per-block conditions in the K loop (as wave-uniform branches around single MFMAs they cut the loop into forty basic
// blocks with a conservative wait in each).
template <int NPROD, bool FULL>
__global__ __launch_bounds__(kDtThreads) void gemm_dt_kernel(DtArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char dt_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    // block → (slab, half): blocks b and b + 8 land on the same XCD (round-robin dispatch), so the halves of a slab are 8 apart
    int slab, half;
    if (g.halves == 2) { const int r = blockIdx.x & 15; half = r >> 3; slab = (blockIdx.x >> 4) * 8 + (r & 7); }
    else { half = 0; slab = blockIdx.x; }
    if (slab >= g.slabs) return;
    const int row_begin = slab * g.slab_rows;
    const int row_end = (row_begin + g.slab_rows) < g.rows ? (row_begin + g.slab_rows) : g.rows;
    const int T = (row_end - row_begin + kDtTileRows - 1) / kDtTileRows;
    const int gb0 = kDtHalfGB * half;                                     // first dx group of this half
    const int gbh = (g.gb - gb0) < kDtHalfGB ? (g.gb - gb0) : kDtHalfGB;       // >= 1 (the launcher's grid)
    // (FULL: five phrase groups + two dx groups, a compile-time constant — every LDS address below is then a lane base plus
    //  an immediate offset; with a run-time G the compiler kept a register per fragment address and spilled)
    const int ga = FULL ? kDtMaxGA : g.ga;
    const int G = FULL ? kDtMaxGA + kDtHalfGB : g.ga + gbh;
    const unsigned tile_bytes = 6u * G * 1024u;            // an image: [k half][plane][group] chunks of 1 KB
    const unsigned plane_bytes = static_cast<unsigned>(G) * 1024u;
    // LDS byte addresses fit 32 bits; the image base as an integer
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)dt_lds));

    // ---- staging: a tile is 16 rows x (M / 4 float4s of phrase | this half's columns / 4 of dx). Wave-slot ws = w + 8 j takes
    // 64 consecutive float4s of ONE operand (phrase: slots 0 .. na - 1, dx: the nb behind them), so that the operand's base
    // and leading dimension are scalars and a lane carries three words per slot: byte offset in the tile, LDS offset, row ----
    const int n_half0 = 128 * half;
    const int ncols_b = (g.N - n_half0) < 128 ? (g.N - n_half0) : 128;
    const int a4 = g.M >> 2, b4 = ncols_b >> 2;
    const int na = (kDtTileRows * a4 + 63) >> 6, nbs = (kDtTileRows * b4 + 63) >> 6;      // na + nbs <= 8 kDtStage (the launcher's check)
    const unsigned dump = 2u * tile_bytes;                 // LDS words nobody reads: the stores of lanes without a float4
    const float* sbase[kDtStage]; int sld[kDtStage];       // wave-uniform
    unsigned soff[kDtStage], sdst[kDtStage]; int srow[kDtStage];
#pragma unroll
    for (int j = 0; j < kDtStage; ++j) {
        const int ws = w + kDtWaves * j;
        const bool is_a = ws < na;
        const int per_row = is_a ? a4 : b4;
        const int idx = (is_a ? ws : ws - na) * 64 + lane;
        const bool ok = ws < na + nbs && idx < kDtTileRows * per_row;
        const int r = ok ? idx / per_row : 0;
        const int col = ok ? 4 * (idx - r * per_row) : 0;                  // inside the operand (dx: inside this half)
        sbase[j] = is_a ? g.A : g.B + n_half0;
        sld[j] = is_a ? g.lda : g.ldb;
        soff[j] = static_cast<unsigned>((r * sld[j] + col) * 4);
        srow[j] = ok ? r : (1 << 24);                                      // (never inside the slab: staged as zeros, into the dump)
        const int grp = is_a ? (col >> 6) : ga + (col >> 6);
        const int k8 = r & 7, ksub = r >> 3, c = (col & 63) >> 3, sub = (col & 7) >> 2;
        sdst[j] = ok ? static_cast<unsigned>(((ksub * 3) * G + grp) * 1024 + (8 * k8 + (c ^ (((k8 >> 1) & 1) << 2))) * 16 + sub * 8) : dump;
    }
    // tiles that lie inside the slab are fetched and cut as they are; the slab's last, partial tile and the tiles past it (the
    // loop requests two ahead) take the rows that exist — the slab's last row again for the others — and stage zeros for the rest
    const int t_whole = (row_end - row_begin) / kDtTileRows;
    // (WHOLE: a compile-time promise that tile t lies inside the slab. The steady-state loop below is built from these
    //  branch-free forms only: with a branch around the loads the compiler's wait counts came out as vmcnt(0) in front of the
    //  loads just issued — the two-tiles-ahead prefetch gone, 3 500 cycles of staging per tile.)
    auto load1 = [&](auto whole_c, int t, u32x4 (&raw)[kDtStage], auto jc) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        constexpr int j = decltype(jc)::value;
        if constexpr (WHOLE) {
            const unsigned char* base = reinterpret_cast<const unsigned char*>(sbase[j] + static_cast<size_t>(row_begin + kDtTileRows * t) * sld[j]);
            raw[j] = *reinterpret_cast<const u32x4*>(base + soff[j]);
        } else {
            int row = row_begin + kDtTileRows * t + (srow[j] & 15);
            row = row < row_end ? row : row_end - 1;
            const unsigned col4 = soff[j] - static_cast<unsigned>((srow[j] & 15) * sld[j] * 4);
            raw[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(sbase[j] + static_cast<size_t>(row) * sld[j]) + col4);
        }
    };
    auto load_tile = [&](auto whole_c, int t, u32x4 (&raw)[kDtStage]) __attribute__((always_inline)) {
        dt_static_for<0, kDtStage>([&](auto jc) __attribute__((always_inline)) { load1(whole_c, t, raw, jc); });
    };
    // float4 j of tile t is cut in two halves (pair 0: elements 0, 1 → the planes' low words; pair 1: elements 2, 3 → the high
    // words) and stored when the second is through: cut_pair(…, 2 j) then cut_pair(…, 2 j + 1)
    unsigned ch[kDtStage], cm[kDtStage], cl[kDtStage];
    auto cut_pair = [&](auto whole_c, int t, const u32x4 (&raw)[kDtStage], int slot, auto kc) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        constexpr int k = decltype(kc)::value, j = k >> 1;
        const bool live = WHOLE || row_begin + kDtTileRows * t + srow[j] < row_end;
        const float x0 = live ? __uint_as_float(raw[j][2 * (k & 1)]) : 0.f, x1 = live ? __uint_as_float(raw[j][2 * (k & 1) + 1]) : 0.f;
        unsigned h, m, l;
        dt_split_pair(x0, x1, h, m, l);
        if constexpr ((k & 1) == 0) { ch[j] = h; cm[j] = m; cl[j] = l; return; }
        // (lanes without a float4 store into the dump — at lds0, not inside an image)
        const unsigned a = (sdst[j] == dump ? lds0 : lds0 + static_cast<unsigned>(slot) * tile_bytes) + sdst[j];
        *(u32x2 __attribute__((address_space(3)))*)(uintptr_t)a = u32x2{ch[j], h};
        *(u32x2 __attribute__((address_space(3)))*)(uintptr_t)(a + plane_bytes) = u32x2{cm[j], m};
        *(u32x2 __attribute__((address_space(3)))*)(uintptr_t)(a + 2 * plane_bytes) = u32x2{cl[j], l};
    };

    // ---- fragments: ds_read_b64_tr_b16. Supplier lane (i, gq, kh): row r = i / 4 (+ 4 per read) of k half kh, the 8 B at
    // piece 4 h + 2 gq + (i % 4) / 2, half (i & 1) — h = which 32 of the group's 64 columns; slot = piece ^ 4 (r >> 1 & 1) ----
    const int i16 = lane & 15, gq = (lane >> 4) & 1, kh = lane >> 5;
    const int fr = i16 >> 2, hx = (fr >> 1) & 1;
    const unsigned flane = static_cast<unsigned>(kh * 3 * G * 1024 + fr * 128 + (2 * gq + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8);
    const unsigned fh[2] = {flane + static_cast<unsigned>(hx * 64), flane + static_cast<unsigned>((1 - hx) * 64)};
    auto frag = [&](unsigned base, unsigned off) __attribute__((always_inline)) -> bf16x8 {
        const unsigned a = base + off;
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(uintptr_t)a);
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(uintptr_t)(a + 512));
        const i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    // this wave's blocks: rows of the output = phrase columns 32 (2 rb + wm) — interleaved, so that block rb of either wave
    // sits in column group rb, half wm: a compile-time chunk and a per-wave lane base —; columns = dx columns 128 half + 32 wn
    const int mblocks = (g.M + 31) / 32;
    const int nrb = (mblocks - wm + 1) / 2 < kDtRB ? (mblocks - wm + 1) / 2 : kDtRB;      // blocks wm, wm + 2, ... < mblocks
    const int n0 = n_half0 + 32 * wn;
    const bool has_cb = n0 < g.N;
    // (blocks the wave does not have: the reads stay inside the image and nothing is multiplied. Columns of the image nobody
    //  stages — phrase columns M .. 32 mblocks, dx columns past N — hold whatever the LDS held: they only ever reach
    //  accumulator rows / columns that are not stored.)
    const unsigned a_lane = fh[wm];
    const unsigned b_lane = fh[has_cb ? (wn & 1) : 0] + static_cast<unsigned>((ga + (has_cb ? (wn >> 1) : 0)) * 1024);

    f32x16 acc[kDtRB];
#pragma unroll
    for (int rb = 0; rb < kDtRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

    // ---- the K loop: the two waves of a SIMD in opposite phases -------------------------------------------------------------
    // Waves w and w + 4 of a workgroup share a SIMD, i.e. a matrix pipe. With all eight waves in step — barrier, staging,
    // fragment reads, MFMAs — that pipe idles while both its waves stage and read, and then both queue for it (4 500 cycles per
    // tile for 1 920 of MFMAs in the first form of this loop; tools/exp/dt_ticks.py). So between two barriers the first half
    // of the workgroup (waves 0-3, one per SIMD) goes stage → read → MULTIPLY and the second half MULTIPLY → stage → read:
    // whenever one wave of a SIMD multiplies, its partner is busy with VALU and LDS. The second half multiplies from the
    // registers it filled in the interval before (its fragments of tile u - 1), so it trails by half an interval, and it runs
    // at s_setprio 1: where the two compete for the matrix pipe the half whose MFMAs come FIRST in its interval must win, or
    // its tail queues behind the partner's whole multiply phase (age-ordered arbitration: 52.6 -> 45.7 us on the planes form).
    //   interval u:  every wave fetches its float4s of tile u + 2 (registers) and cuts + stores those of tile u + 1 (fetched
    //                during interval u - 1) into the other LDS image;
    //                half 0: fragments of tile u → registers, multiply tile u
    //                half 1: multiply tile u - 1 (registers), fragments of tile u → registers
    //   then the barrier: behind it tile u + 1 is in LDS for everybody and nobody reads tile u's image any more.
    bf16x8 bh, bm, bl, ah[kDtRB], am[kDtRB], al[kDtRB];
    auto read_tile = [&](int slot) __attribute__((always_inline)) {
        const unsigned img = lds0 + static_cast<unsigned>(slot) * tile_bytes;
        bh = frag(img + b_lane, 0); bm = frag(img + b_lane, plane_bytes); bl = frag(img + b_lane, 2 * plane_bytes);
#pragma unroll
        for (int rb = 0; rb < kDtRB; ++rb) {
            const unsigned off = static_cast<unsigned>((rb < kDtMaxGA ? rb : 0) * 1024);      // (group rb; a wave without block rb reads it all the same)
            ah[rb] = frag(img + a_lane, off); am[rb] = frag(img + a_lane, plane_bytes + off); al[rb] = frag(img + a_lane, 2 * plane_bytes + off);
        }
    };
    // PRODUCT-major over the row blocks: consecutive MFMAs go to different accumulators, so that the wave keeps the matrix pipe
    // busy on its own (six dependent MFMAs in a row per block ran at the pipe's latency, not its rate: 60-90 cycles per MFMA
    // measured). Smallest products first; operands fed swapped: the accumulator's rows run along the columns of dx, four
    // consecutive ones per register quad (16 B stores).
    // STAGE: the wave's staging work of the interval — the four fetches of tile t + 1, the eight pair cuts of tile t and their
    // stores — rides BETWEEN its own MFMAs, a pair cut (≈ 13 instructions) behind every fourth: a SIMD issues a handful of other
    // instructions in the shadow of each MFMA of the SAME wave, but a wave that stages while its partner streams MFMAs gets
    // next to nothing (measured: four global loads took 1 400 cycles to issue, thirty-six LDS reads as long, beside a partner's
    // multiply phase; at any s_setprio). The scheduling barriers pin the interleave.
    auto mfma1 = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int rnd = i / kDtRB + (NPROD == 9 ? 0 : 3), rb = i % kDtRB;
        // rounds: (bl,al) (bl,am) (bm,al) | (bl,ah) (bh,al) (bm,am) (bm,ah) (bh,am) (bh,ah)
        if (!(FULL || rb < nrb)) return;
        if constexpr (rnd == 0) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, al[rb], acc[rb], 0, 0, 0);
        else if constexpr (rnd == 1) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, am[rb], acc[rb], 0, 0, 0);
        else if constexpr (rnd == 2) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, al[rb], acc[rb], 0, 0, 0);
        else if constexpr (rnd == 3) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[rb], acc[rb], 0, 0, 0);
        else if constexpr (rnd == 4) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[rb], acc[rb], 0, 0, 0);
        else if constexpr (rnd == 5) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, am[rb], acc[rb], 0, 0, 0);
        else if constexpr (rnd == 6) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, ah[rb], acc[rb], 0, 0, 0);
        else if constexpr (rnd == 7) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, am[rb], acc[rb], 0, 0, 0);
        else acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[rb], acc[rb], 0, 0, 0);
    };
    constexpr int kMfmas = NPROD * kDtRB;
    auto multiply = [&]() __attribute__((always_inline)) { dt_static_for<0, kMfmas>([&](auto ic) __attribute__((always_inline)) { mfma1(ic); }); };
    // ONE register set for the fetched float4s: float4 j of tile t_cut + 1 is requested right behind the cut of float4 j of tile
    // t_cut, into the registers that cut has just freed — every fetch has exactly one interval to arrive (two sets, fetched two
    // tiles ahead, put the kernel past 256 registers: the second half of the workgroup carries its 72 fragment registers across
    // the barrier and spilled 70 times per tile pair).
    auto multiply_and_stage = [&](auto wc, int t_cut, u32x4 (&raw)[kDtStage], int cut_slot) __attribute__((always_inline)) {
        constexpr int kCuts = 2 * kDtStage;
        dt_static_for<0, kMfmas>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            mfma1(ic);
            // pair cut k behind MFMA (k + 1) kMfmas / 9 - 1: eight of them, evenly spread, the last MFMAs without
            dt_static_for<0, kCuts>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if constexpr (i == (k + 1) * kMfmas / (kCuts + 1) - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    cut_pair(wc, t_cut, raw, cut_slot, kc);
                    if constexpr (k & 1) load1(wc, t_cut + 1, raw, std::integral_constant<int, (k >> 1)>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    };
    // the same without MFMAs (the second half's first interval)
    auto stage = [&](auto wc, int t_cut, u32x4 (&raw)[kDtStage], int cut_slot) __attribute__((always_inline)) {
        dt_static_for<0, 2 * kDtStage>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            cut_pair(wc, t_cut, raw, cut_slot, kc);
            if constexpr (k & 1) load1(wc, t_cut + 1, raw, std::integral_constant<int, (k >> 1)>{});
        });
    };
    // every LDS operation of this wave has been carried out (the compiler does not see the raw barrier as one that needs it)
    auto meet = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    u32x4 raw[kDtStage];
    constexpr std::false_type any_c{};
    constexpr std::true_type whole_c{};
    load_tile(any_c, 0, raw);
    stage(any_c, 0, raw, 0);               // tile 0 → image 0; tile 1 requested
    if (w >= 4) __builtin_amdgcn_s_setprio(1);      // (55.2 against 57.7 us without; the first half prioritised: no different from none)
    meet();
    // intervals u < u_whole touch whole tiles only (they stage tile u + 1 and request tile u + 2)
    const int u_whole = t_whole >= 3 ? t_whole - 2 : 0;
    if (w < 4) {
        auto interval = [&](auto wc, int u) __attribute__((always_inline)) {
            const int slot = u & 1;
            DT_TICK(u, 0);
            read_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            DT_TICK(u, 1);
            multiply_and_stage(wc, u + 1, raw, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            DT_TICK(u, 2);
            meet();
            DT_TICK(u, 3);
        };
        int u = 0;
        for (; u < u_whole; ++u) interval(whole_c, u);
        for (; u < T; ++u) interval(any_c, u);
    } else {
        auto interval = [&](auto wc, int u) __attribute__((always_inline)) {
            const int slot = u & 1;
            DT_TICK(u, 0);
            multiply_and_stage(wc, u + 1, raw, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            DT_TICK(u, 1);
            read_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            DT_TICK(u, 2);
            meet();
            DT_TICK(u, 3);
        };
        // (interval 0 has nothing to multiply yet)
        int u = 1;
        if (T > 0) { stage(any_c, 1, raw, 1); read_tile(0); meet(); }
        for (; u < u_whole; ++u) interval(whole_c, u);
        for (; u < T; ++u) interval(any_c, u);
        if (T > 0) multiply();
    }

    // ---- this slab's partial: acc[rb][4 q + r] = P[slab][32 (2 rb + wm) + i][n0 + 8 q + 4 kh + r] ----
    float* P = g.P + static_cast<size_t>(slab) * g.p_stride;
    const int i32 = lane & 31;
#pragma unroll
    for (int rb = 0; rb < kDtRB; ++rb) {
        if (rb < nrb && has_cb) {
            const int m = 32 * (2 * rb + wm) + i32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 8 * q + 4 * kh;
                if (m < g.M && n < g.N)
                    *reinterpret_cast<f32x4*>(P + static_cast<size_t>(m) * g.ldc + n) =
                        f32x4{acc[rb][4 * q], acc[rb][4 * q + 1], acc[rb][4 * q + 2], acc[rb][4 * q + 3]};
            }
        }
    }
}

// ---- the product --------------------------------------------------------------------------------------------------------
bool gemm_dt_covers(int M, int N, int rows) {
    return gemm_split_products() != 0 && M % 4 == 0 && N % 4 == 0 && M >= 4 && M <= 64 * kDtMaxGA && N >= 4 && N <= 64 * kDtMaxGB && rows >= 1;
}
static int dt_slab_rows(int rows, int want) {
    if (want < 1) want = 1;
    const int len = (rows + want - 1) / want;
    return ((len + kDtTileRows - 1) / kDtTileRows) * kDtTileRows;
}
// slabs the product will use for `want` (slab lengths are multiples of 16 rows): never more than `want`
int gemm_dt_slabs(int rows, int want) {
    const int len = dt_slab_rows(rows, want);
    return (rows + len - 1) / len;
}
// How many slabs a batch is cut into by default: two workgroups per slab, a workgroup per CU, from 64 rows per slab on (four K
// tiles: below that a workgroup is prologue and epilogue — 0.15 MB of partials each — and little else)
int gemm_dt_default_slabs(int rows, int cus) {
    int want = rows / 64;
    want = want < 1 ? 1 : want;
    return want < cus / 2 ? want : cus / 2;
}

// A [rows][M] (lda), B [rows][N] (ldb); partial [slabs][M][N] (ldc = N); the caller adds the slabs (launch_splitk_reduce).
// false: shape not covered, nothing launched.
bool launch_gemm_dt(const float* A, const float* B, float* partial, int M, int N, int rows, int lda, int ldb, int want_slabs, hipStream_t s) {
    if (!gemm_dt_covers(M, N, rows) || (lda % 4) || (ldb % 4) || lda < M || ldb < N) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(partial)) % 16) return false;
    const int nprod = gemm_split_products();
    DtArgs g{};
    g.A = A; g.B = B; g.lda = lda; g.ldb = ldb;
    g.P = partial; g.p_stride = static_cast<size_t>(M) * N; g.ldc = N;
    g.rows = rows; g.M = M; g.N = N; g.slab_rows = dt_slab_rows(rows, want_slabs);
    g.ga = (M + 63) / 64; g.gb = (N + 63) / 64;
    g.halves = g.gb > kDtHalfGB ? 2 : 1;
    const int slabs = gemm_dt_slabs(rows, want_slabs);
    g.slabs = slabs;
    const int grid = g.halves == 2 ? 16 * ((slabs + 7) / 8) : slabs;
    const int gw = g.ga + (g.gb < kDtHalfGB ? g.gb : kDtHalfGB);
    const size_t lds = static_cast<size_t>(2) * 6 * gw * 1024 + 2 * static_cast<size_t>(gw) * 1024 + 1024;      // two images | dump (three planes' offsets)
    const bool full = M > 288 && N > 224;
    static std::atomic<bool> attr_set[kDtMaxDevices][4];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDtMaxDevices) return false;
    const int which = (nprod == 9 ? 1 : 0) + (full ? 2 : 0);
    const void* fns[4] = {reinterpret_cast<const void*>(&gemm_dt_kernel<6, false>), reinterpret_cast<const void*>(&gemm_dt_kernel<9, false>),
                          reinterpret_cast<const void*>(&gemm_dt_kernel<6, true>), reinterpret_cast<const void*>(&gemm_dt_kernel<9, true>)};
    if (!attr_set[dev][which].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 6 + 2) * (kDtMaxGA + kDtHalfGB) * 1024 + 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev][which].store(true, std::memory_order_release);
    }
    (void)hipGetLastError();
    switch (which) {
        case 0: NVSM_LAUNCH((gemm_dt_kernel<6, false>), dim3(grid), dim3(kDtThreads), lds, s, g); break;
        case 1: NVSM_LAUNCH((gemm_dt_kernel<9, false>), dim3(grid), dim3(kDtThreads), lds, s, g); break;
        case 2: NVSM_LAUNCH((gemm_dt_kernel<6, true>), dim3(grid), dim3(kDtThreads), lds, s, g); break;
        default: NVSM_LAUNCH((gemm_dt_kernel<9, true>), dim3(grid), dim3(kDtThreads), lds, s, g); break;
    }
    return hipGetLastError() == hipSuccess;
}

}  // namespace cunvsm

#ifdef NVSM_DT_TIMING
extern "C" int nvsm_debug_dt_ticks(unsigned long long* out, int n) {
    (void)hipDeviceSynchronize();
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(cunvsm::g_dt_ticks), sizeof(unsigned long long) * static_cast<size_t>(n)));
}
#endif
