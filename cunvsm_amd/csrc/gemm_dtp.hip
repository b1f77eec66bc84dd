// dT[d_w][d_e] = phraseᵀ[d_w x B] · dx[B x d_e], the projection gradient (cpp/params.cu:526-531), on the bf16 matrix pipe at
// fp32 accuracy (the split-bf16 arithmetic of gemm_split.hip: x = h + m + l exactly, six or nine bf16 MFMAs per product, fp32
// accumulation) — from operands that ARRIVE as bf16 planes. Both operands of this product are batch-sized and enter it with
// the batch as the reduction dimension, i.e. transposed against the way their producers write them; round 3's kernel
// (gemm_dt.hip) fetched fp32, cut every element into its planes and transposed in registers: ≈ 900 staging instructions per
// SIMD and K tile against 120 MFMAs (MfmaUtil 17 %). Here
//   * the producers write the planes: the forward / backward projection products cut their A operand (phrase / dx) on its way
//     into LDS anyway and store the pieces (gemm_split.hip, `planes_out`); below their batch sizes launch_cut_planes does it.
//     Layout (kernels.h Planes): three row-major bf16 matrices [rows][pitch], pitch a multiple of 64 columns, padding columns
//     zero, one all-zero row behind the last (K tails read it instead of branching);
//   * a workgroup (eight waves) owns a slab of the batch and one HALF of the d_e columns: 320 x 128 of the output. The two
//     workgroups of a slab run on the same XCD at the same time (block index → (slab, half) below), so the phrase planes,
//     which both need, come from HBM once and from that XCD's L2 the second time;
//   * K runs in tiles of 16 batch rows. A tile moves global → LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
//     ds_write, one address computation per KB), as 42 chunks of 1 KB = 8 rows x 128 B, the 16 B pieces of a chunk permuted
//     (piece c of row k sits at slot 8 k + (c ^ 4 (k >> 1 & 1))) by permuting the SOURCE addresses — the destination of an
//     LDS-DMA is lane-linear. Three tile images (3 x 42 KB): two tiles are in flight while one is multiplied;
//   * the MFMA fragments, whose K runs along the batch, come out of that row-major image with ds_read_b64_tr_b16 (the 4 x 4
//     transposition happens in the LDS read path); with the permutation above the 4 rows x 64 B a half-wave touches fall on
//     all 64 banks once;
//   * wave (wm, wn) of a 2 x 4 grid owns 160 x 32 of the output: 5 blocks of v_mfma_f32_32x32x16_bf16, 80 accumulator
//     registers; one barrier per tile: [tile t landed, everybody through with tile t - 1] → request tile t + 2 → multiply tile t;
//   * the slab's partial product goes to partial[slab] and launch_splitk_reduce adds the slabs in order (deterministic).
// What bounds it: 47 GFLOP of bf16 MFMAs (19 us at the 2.5 PFLOP/s peak) against 177 MB of planes + 39 MB of partials
// (≈ 40 us at 5.5 TB/s): HBM / MALL, as everything else in the step's back half.
#include "../../include/cunvsm_amd.h"
#include "kernels.h"
#include "device_utils.h"

#include <atomic>
#include <cstdlib>

namespace cunvsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));

constexpr int kDtpWaves = 8, kDtpThreads = kDtpWaves * 64;
constexpr int kDtpMaxGA = 5, kDtpMaxGB = 4;            // 64-column groups of the two operands: M <= 320, N <= 256
constexpr int kDtpHalfGB = 2;                          // ... of dx per workgroup: 128 columns
constexpr int kDtpTileRows = 16;                       // batch rows per K tile = one MFMA k step
constexpr int kDtpRing = 3;                            // tile images in LDS
constexpr int kDtpIssue = 6;                           // LDS-DMA instructions per wave and tile, at most: ceil(6 x 7 chunks / 8 waves)
constexpr int kDtpRB = 5;                              // 32 x 32 blocks per wave: 5 along the rows of the output (phrase columns) x 1
constexpr int kDtpMaxDevices = 64;

struct DtpArgs {
    const unsigned char* A; const unsigned char* B;    // plane h of phrase [rows][a_pitch] / dx [rows][b_pitch] (bf16)
    unsigned a_plane, b_plane;                         // bytes from one plane to the next
    unsigned a_pitch, b_pitch;                         // bytes per row
    int a_zero_row, b_zero_row;                        // the all-zero row of each
    float* P; size_t p_stride; int ldc;                // partial [slabs][M][N]
    int rows, M, N, slab_rows, slabs;                  // rows = batch (the reduction); slab_rows a multiple of 16
    int ga, gb;                                        // 64-column groups in use
    int halves;                                        // 1: N <= 128
    int prio;                                          // (experiments) 0: none, 1: s_setprio 1 for waves 4-7, 2: for waves 0-3
};

#ifdef NVSM_DT_TIMING
// experiments (make dbg; tools/exp/dtp_ticks.py): shader-clock stamps of waves 0, 4 (one SIMD), 1 and 7 of two workgroups inside the K loop
__device__ unsigned long long g_dtp_ticks[2 * 4 * 16 * 8];
#define DTP_TICK(t, point) do { const int _bw = blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 1 : -1); \
    const int _ww = w == 0 ? 0 : (w == 4 ? 1 : (w == 1 ? 2 : (w == 7 ? 3 : -1))); \
    if (_bw >= 0 && _ww >= 0 && lane == 0 && (t) < 16) g_dtp_ticks[((_bw * 4 + _ww) * 16 + (t)) * 8 + (point)] = clock64(); } while (0)
#else
#define DTP_TICK(t, point) do {} while (0)
#endif

template <int C> __device__ __forceinline__ void dtp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C) : "memory"); }
// all but the `c` most recently issued memory operations of this wave have completed (c wave-uniform)
__device__ __forceinline__ void dtp_wait_vm_dyn(int c) {
    switch (c) {
        case 0: dtp_wait_vm<0>(); break;
        case 1: dtp_wait_vm<1>(); break;
        case 2: dtp_wait_vm<2>(); break;
        case 3: dtp_wait_vm<3>(); break;
        case 4: dtp_wait_vm<4>(); break;
        case 5: dtp_wait_vm<5>(); break;
        default: dtp_wait_vm<6>(); break;
    }
}

// FULL: every wave has all its blocks (ten row blocks, eight column blocks: 289 <= M <= 320, 225 <= N <= 256 — the projection's
// shape): no per-block conditions in the K loop (as wave-uniform branches around single MFMAs they cut the loop into forty basic
// blocks with a conservative wait in each).
template <int NPROD, bool FULL>
__global__ __launch_bounds__(kDtpThreads) void gemm_dtp_kernel(DtpArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char dtp_lds[];
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
    const int T = (row_end - row_begin + kDtpTileRows - 1) / kDtpTileRows;
    const int gb0 = kDtpHalfGB * half;                                    // first dx group of this half
    const int gbh = (g.gb - gb0) < kDtpHalfGB ? (g.gb - gb0) : kDtpHalfGB;      // >= 1 (the launcher's grid)
    const int G = g.ga + gbh;
    const unsigned tile_bytes = 6u * G * 1024u;            // [k half][plane][group] chunks of 1 KB

    // LDS byte addresses fit 32 bits; the image base as an integer
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)dtp_lds));
    // ---- LDS-DMA: chunk q = w + 8 j of a tile ----
    // lane: row k = lane / 8 of the chunk's eight, 16 B piece c' = lane % 8 of the SLOT order, i.e. source piece c' ^ 4 (k >> 1 & 1)
    const int dk = lane >> 3;
    const unsigned dpiece = static_cast<unsigned>(((lane & 7) ^ (((dk >> 1) & 1) << 2)) * 16);
    const int nissue = FULL ? (w < 2 ? 6 : 5) : (6 * G - w + 7) / 8;      // this wave's chunks per tile (wave-uniform; <= kDtpIssue)
    const unsigned char* dbase[kDtpIssue];                 // wave-uniform: plane + column group
    unsigned dpitch[kDtpIssue]; int dzero[kDtpIssue], dksub[kDtpIssue]; unsigned dlds[kDtpIssue];
#pragma unroll
    for (int j = 0; j < kDtpIssue; ++j) {
        int q = w + 8 * j;
        q = q < 6 * G ? q : 6 * G - 1;
        const int ksub = q / (3 * G), r = q - ksub * 3 * G, plane = r / G, grp = r - plane * G;
        const bool is_a = grp < g.ga;
        dbase[j] = is_a ? g.A + static_cast<size_t>(plane) * g.a_plane + grp * 128
                        : g.B + static_cast<size_t>(plane) * g.b_plane + (gb0 + grp - g.ga) * 128;
        dpitch[j] = is_a ? g.a_pitch : g.b_pitch;
        dzero[j] = is_a ? g.a_zero_row : g.b_zero_row;
        dksub[j] = ksub;
        dlds[j] = static_cast<unsigned>(q) * 1024u;
    }
    // (inline asm: hipcc waits for EVERY LDS-DMA in flight — vmcnt(0) — in front of the next LDS read it sees, which would drain
    //  the requests for tiles t + 1 and t + 2 before tile t is multiplied. Issued this way the compiler knows nothing of them; the
    //  waits are the counted `s_waitcnt vmcnt` below, each followed by the barrier that orders the data for the other waves'
    //  reads. M0 — the LDS-DMA destination base — is written in the statement that uses it and restored.)
    auto issue1 = [&](int j, int t, int slot) {
        if ((FULL && j < 5) || j < nissue) {
            int row = row_begin + kDtpTileRows * t + 8 * dksub[j] + dk;
            row = row < row_end ? row : dzero[j];
            const unsigned voff = static_cast<unsigned>(row) * dpitch[j] + dpiece;
            const unsigned dst = lds0 + static_cast<unsigned>(slot) * tile_bytes + dlds[j];
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(dst), "s"(dbase[j]) : "memory");
        }
    };
    auto issue = [&](int t, int slot) {
#pragma unroll
        for (int j = 0; j < kDtpIssue; ++j) issue1(j, t, slot);
    };

    // ---- fragments: ds_read_b64_tr_b16. Supplier lane (i, gq, kh): row r = i / 4 (+ 4 per read) of k half kh, the 8 B at
    // piece 4 h + 2 gq + (i % 4) / 2, half (i & 1) — h = which 32 of the group's 64 columns; slot = piece ^ 4 (r >> 1 & 1) ----
    const int i16 = lane & 15, gq = (lane >> 4) & 1, kh = lane >> 5;
    const int fr = i16 >> 2, hx = (fr >> 1) & 1;
    const unsigned flane = static_cast<unsigned>(kh * 3 * G * 1024 + fr * 128 + (2 * gq + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8);
    const unsigned fh[2] = {flane + static_cast<unsigned>(hx * 64), flane + static_cast<unsigned>((1 - hx) * 64)};
    auto frag = [&](unsigned img, int plane, int grp, int h) -> bf16x8 {
        const unsigned a = img + fh[h] + static_cast<unsigned>((plane * G + grp) * 1024);
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(uintptr_t)a);
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(uintptr_t)(a + 512));
        const i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    // this wave's blocks: rows of the output = phrase columns 32 (5 wm + rb); columns = dx columns 128 half + 32 wn
    const int mblocks = (g.M + 31) / 32;
    const int rb0 = kDtpRB * wm;
    const int nrb = (mblocks - rb0) < kDtpRB ? (mblocks - rb0 > 0 ? mblocks - rb0 : 0) : kDtpRB;
    const int n0 = 128 * half + 32 * wn;
    const bool has_cb = n0 < g.N;
    // (blocks the wave does not have: the reads stay inside the image — block 0 of the operand — and nothing is multiplied)
    int a_grp[kDtpRB], a_half[kDtpRB];
#pragma unroll
    for (int rb = 0; rb < kDtpRB; ++rb) { const int mb = rb < nrb ? rb0 + rb : 0; a_grp[rb] = mb >> 1; a_half[rb] = mb & 1; }
    const int b_grp = g.ga + (has_cb ? (wn >> 1) : 0), b_half = has_cb ? (wn & 1) : 0;

    f32x16 acc[kDtpRB];
#pragma unroll
    for (int rb = 0; rb < kDtpRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

    // ---- the K loop: the two waves of a SIMD in opposite phases -------------------------------------------------------------
    // Waves w and w + 4 of a workgroup share a SIMD, i.e. a matrix pipe. With all eight waves in step — barrier, fragment reads,
    // requests, MFMAs — that pipe idled while both its waves read and both then queued for it: 4 500 cycles per tile for 1 920
    // of MFMAs (tools/exp/dtp_ticks.py). So between two barriers the first half of the workgroup (waves 0-3, one per SIMD) goes
    // read → request → MULTIPLY and the second half MULTIPLY → read → request: whenever one wave of a SIMD multiplies, its partner
    // is busy with LDS and the DMA queue. The second half multiplies from registers it filled in the interval before (its
    // fragments of tile u - 1), so it trails by half an interval; one barrier per tile still does:
    //   interval u:  half 0: fragments of tile u → registers, request tile u + 2, multiply tile u
    //                half 1: multiply tile u - 1 (registers), fragments of tile u → registers, request tile u + 2
    //   then every wave waits for its share of tile u + 1 (the request before last) and meets the others.
    // Behind that barrier tile u + 1 has landed for everybody and nobody reads tile u any more: its image takes tile u + 3 in
    // the next interval. Tiles past the last one are requested all the same (rows >= row_end: the zero row; never read), so
    // that every interval finds the same number of requests in flight.
    bf16x8 bh, bm, bl, ah[kDtpRB], am[kDtpRB], al[kDtpRB];
    auto read_tile = [&](int slot) {
        const unsigned img = lds0 + static_cast<unsigned>(slot) * tile_bytes;
        bh = frag(img, 0, b_grp, b_half); bm = frag(img, 1, b_grp, b_half); bl = frag(img, 2, b_grp, b_half);
#pragma unroll
        for (int rb = 0; rb < kDtpRB; ++rb) {
            ah[rb] = frag(img, 0, a_grp[rb], a_half[rb]); am[rb] = frag(img, 1, a_grp[rb], a_half[rb]); al[rb] = frag(img, 2, a_grp[rb], a_half[rb]);
        }
    };
    // PRODUCT-major over the row blocks: consecutive MFMAs go to different accumulators, so that the wave keeps the matrix pipe
    // busy on its own (six dependent MFMAs in a row per block ran at the pipe's latency, not its rate: 60-90 cycles per MFMA
    // measured). Smallest products first; operands fed swapped: the accumulator's rows run along the columns of dx, four
    // consecutive ones per register quad (16 B stores).
    auto multiply = [&] {
#define DTP_ROUND(x, y)                                                                                                  \
        _Pragma("unroll") for (int rb = 0; rb < kDtpRB; ++rb)                                                           \
            if (FULL || rb < nrb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y[rb], acc[rb], 0, 0, 0)
        if (NPROD == 9) { DTP_ROUND(bl, al); DTP_ROUND(bl, am); DTP_ROUND(bm, al); }
        DTP_ROUND(bl, ah); DTP_ROUND(bh, al); DTP_ROUND(bm, am); DTP_ROUND(bm, ah); DTP_ROUND(bh, am); DTP_ROUND(bh, ah);
#undef DTP_ROUND
    };
    auto wait_share = [&] {
        if constexpr (FULL) { if (w < 2) dtp_wait_vm<6>(); else dtp_wait_vm<5>(); }
        else dtp_wait_vm_dyn(nissue);
    };
    issue(0, 0);
    issue(1, 1);
    wait_share();
    if (g.prio == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    if (g.prio == 2 && w < 4) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_s_barrier();
    if (w < 4) {
        int slot = 0;
        for (int u = 0; u < T; ++u) {
            DTP_TICK(u, 0);
            read_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            DTP_TICK(u, 1);
            issue(u + 2, slot == 0 ? 2 : slot - 1);
            __builtin_amdgcn_sched_barrier(0);
            DTP_TICK(u, 2);
            multiply();
            __builtin_amdgcn_sched_barrier(0);
            DTP_TICK(u, 3);
            wait_share();
            DTP_TICK(u, 4);
            __builtin_amdgcn_s_barrier();
            DTP_TICK(u, 5);
            slot = slot == 2 ? 0 : slot + 1;
        }
    } else {
        int slot = 0;
        for (int u = 0; u < T; ++u) {
            DTP_TICK(u, 0);
            if (u > 0) multiply();
            __builtin_amdgcn_sched_barrier(0);
            DTP_TICK(u, 1);
            read_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            DTP_TICK(u, 2);
            issue(u + 2, slot == 0 ? 2 : slot - 1);
            __builtin_amdgcn_sched_barrier(0);
            DTP_TICK(u, 3);
            wait_share();
            DTP_TICK(u, 4);
            __builtin_amdgcn_s_barrier();
            DTP_TICK(u, 5);
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (T > 0) multiply();
    }
    dtp_wait_vm<0>();      // (the last, unread requests: nothing of this workgroup is in flight when it ends)

    // ---- this slab's partial: acc[rb][4 q + r] = P[slab][32 (rb0 + rb) + i][n0 + 8 q + 4 kh + r] ----
    float* P = g.P + static_cast<size_t>(slab) * g.p_stride;
    const int i32 = lane & 31;
#pragma unroll
    for (int rb = 0; rb < kDtpRB; ++rb) {
        if (rb < nrb && has_cb) {
            const int m = 32 * (rb0 + rb) + i32;
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

// ---- planes (kernels.h Planes) ------------------------------------------------------------------------------------------
int planes_pitch(int cols) { return 64 * ((cols + 63) / 64); }
size_t planes_bytes(int64_t rows_cap, int cols) { return static_cast<size_t>(3) * (rows_cap + 1) * planes_pitch(cols) * 2; }
Planes planes_view(void* buf, int64_t rows_cap, int cols) {
    Planes p{};
    p.p = static_cast<unsigned char*>(buf); p.pitch = planes_pitch(cols); p.cols = cols; p.rows_cap = rows_cap;
    p.plane_bytes = static_cast<size_t>(rows_cap + 1) * p.pitch * 2;
    return p;
}

// x0, x1 -> one 32-bit word per plane (gemm_split.hip split_pair)
__device__ __forceinline__ void dtp_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r = v - f32x2_t{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    const f32x2_t s2 = r - f32x2_t{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(s2, bf16x2_t));
}

// X [rows][cols] (ld) -> its planes; a thread per four columns (padding columns: zeros). The stand-alone producer: batches
// below the split-bf16 products' (their A operands pass through the exact-fp32 kernels uncut), tests, experiments.
__global__ __launch_bounds__(256) void cut_planes_kernel(const float* __restrict__ X, int64_t rows, int cols, int ld, Planes p) {
    const int q4 = p.pitch / 4;
    const int64_t total = rows * q4;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += gridDim.x * 256ll) {
        const int64_t row = idx / q4;
        const int c = static_cast<int>(idx - row * q4) * 4;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < cols && ld % 4 == 0) {
            const float4 v = *reinterpret_cast<const float4*>(X + row * ld + c);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < cols) x[e] = X[row * ld + c + e];
        }
        unsigned h0, m0, l0, h1, m1, l1;
        dtp_split_pair(x[0], x[1], h0, m0, l0);
        dtp_split_pair(x[2], x[3], h1, m1, l1);
        unsigned char* d = p.p + (static_cast<size_t>(row) * p.pitch + c) * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + p.plane_bytes) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(d + 2 * p.plane_bytes) = make_uint2(l0, l1);
    }
}
void launch_cut_planes(const float* X, int64_t rows, int cols, int ld, const Planes& p, hipStream_t s) {
    if (rows <= 0) return;
    if (rows > p.rows_cap || cols != p.cols) throw Error(NVSM_ERR_INVALID_ARGUMENT, "launch_cut_planes: the planes were made for another shape");
    const int64_t total = rows * (p.pitch / 4);
    const int64_t blocks = (total + 255) / 256;
    NVSM_LAUNCH(cut_planes_kernel, dim3(static_cast<unsigned>(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, X, rows, cols, ld, p);
}
// (tests) planes -> fp32: h + m + l is exact
__global__ __launch_bounds__(256) void join_planes_kernel(Planes p, int64_t rows, int cols, float* __restrict__ X) {
    const int64_t total = rows * cols;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += gridDim.x * 256ll) {
        const int64_t row = idx / cols;
        const int c = static_cast<int>(idx - row * cols);
        const unsigned short* d = reinterpret_cast<const unsigned short*>(p.p) + static_cast<size_t>(row) * p.pitch + c;
        const size_t ps = p.plane_bytes / 2;
        X[idx] = (__uint_as_float(static_cast<unsigned>(d[0]) << 16) + __uint_as_float(static_cast<unsigned>(d[ps]) << 16)) +
                 __uint_as_float(static_cast<unsigned>(d[2 * ps]) << 16);
    }
}
void launch_join_planes(const Planes& p, int64_t rows, int cols, float* X, hipStream_t s) {
    if (rows <= 0) return;
    const int64_t blocks = (rows * cols + 255) / 256;
    NVSM_LAUNCH(join_planes_kernel, dim3(static_cast<unsigned>(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, p, rows, cols, X);
}

// ---- the product --------------------------------------------------------------------------------------------------------
bool gemm_dtp_covers(int M, int N, int rows) {
    return gemm_split_products() != 0 && M % 4 == 0 && N % 4 == 0 && M >= 4 && M <= 64 * kDtpMaxGA && N >= 4 && N <= 64 * kDtpMaxGB && rows >= 1;
}
static int dtp_slab_rows(int rows, int want) {
    if (want < 1) want = 1;
    const int len = (rows + want - 1) / want;
    return ((len + kDtpTileRows - 1) / kDtpTileRows) * kDtpTileRows;
}
// slabs the product will use for `want` (slab lengths are multiples of 16 rows): never more than `want`
int gemm_dtp_slabs(int rows, int want) {
    const int len = dtp_slab_rows(rows, want);
    return (rows + len - 1) / len;
}
// How many slabs a batch is cut into by default: two workgroups per slab, a workgroup per CU, from 96 rows per slab on (six K
// tiles: below that a workgroup is prologue and epilogue — 0.15 MB of partials each — and little else)
int gemm_dtp_default_slabs(int rows, int cus) {
    int want = rows / 96;
    want = want < 1 ? 1 : want;
    return want < cus / 2 ? want : cus / 2;
}

// partial [slabs][M][N] (ldc = N); the caller adds the slabs (launch_splitk_reduce). false: shape not covered, nothing launched.
bool launch_gemm_dtp(const Planes& A, const Planes& B, float* partial, int M, int N, int rows, int want_slabs, hipStream_t s) {
    if (!gemm_dtp_covers(M, N, rows) || A.cols != M || B.cols != N || rows > A.rows_cap || rows > B.rows_cap) return false;
    if (reinterpret_cast<uintptr_t>(partial) % 16 || reinterpret_cast<uintptr_t>(A.p) % 16 || reinterpret_cast<uintptr_t>(B.p) % 16) return false;
    if (A.plane_bytes >= (1ull << 32) || B.plane_bytes >= (1ull << 32)) return false;      // 32-bit byte offsets inside a plane
    const int nprod = gemm_split_products();
    DtpArgs g{};
    g.A = A.p; g.B = B.p;
    g.a_plane = static_cast<unsigned>(A.plane_bytes); g.b_plane = static_cast<unsigned>(B.plane_bytes);
    g.a_pitch = static_cast<unsigned>(A.pitch * 2); g.b_pitch = static_cast<unsigned>(B.pitch * 2);
    g.a_zero_row = static_cast<int>(A.rows_cap); g.b_zero_row = static_cast<int>(B.rows_cap);
    g.P = partial; g.p_stride = static_cast<size_t>(M) * N; g.ldc = N;
    g.rows = rows; g.M = M; g.N = N; g.slab_rows = dtp_slab_rows(rows, want_slabs);
    g.ga = A.pitch / 64; g.gb = B.pitch / 64;
    g.halves = g.gb > kDtpHalfGB ? 2 : 1;
    { const char* e = std::getenv("NVSM_DTP_PRIO"); g.prio = e ? std::atoi(e) : 1; }
    const int slabs = gemm_dtp_slabs(rows, want_slabs);
    g.slabs = slabs;
    const int grid = g.halves == 2 ? 16 * ((slabs + 7) / 8) : slabs;
    const size_t lds = static_cast<size_t>(kDtpRing) * 6 * (g.ga + (g.gb < kDtpHalfGB ? g.gb : kDtpHalfGB)) * 1024;
    const bool full = M > 288 && N > 224;
    static std::atomic<bool> attr_set[kDtpMaxDevices][4];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDtpMaxDevices) return false;
    const int which = (nprod == 9 ? 1 : 0) + (full ? 2 : 0);
    const void* fns[4] = {reinterpret_cast<const void*>(&gemm_dtp_kernel<6, false>), reinterpret_cast<const void*>(&gemm_dtp_kernel<9, false>),
                          reinterpret_cast<const void*>(&gemm_dtp_kernel<6, true>), reinterpret_cast<const void*>(&gemm_dtp_kernel<9, true>)};
    if (!attr_set[dev][which].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, kDtpRing * 6 * (kDtpMaxGA + kDtpHalfGB) * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev][which].store(true, std::memory_order_release);
    }
    (void)hipGetLastError();
    switch (which) {
        case 0: NVSM_LAUNCH((gemm_dtp_kernel<6, false>), dim3(grid), dim3(kDtpThreads), lds, s, g); break;
        case 1: NVSM_LAUNCH((gemm_dtp_kernel<9, false>), dim3(grid), dim3(kDtpThreads), lds, s, g); break;
        case 2: NVSM_LAUNCH((gemm_dtp_kernel<6, true>), dim3(grid), dim3(kDtpThreads), lds, s, g); break;
        default: NVSM_LAUNCH((gemm_dtp_kernel<9, true>), dim3(grid), dim3(kDtpThreads), lds, s, g); break;
    }
    return hipGetLastError() == hipSuccess;
}

}  // namespace cunvsm

#ifdef NVSM_DT_TIMING
extern "C" int nvsm_debug_dtp_ticks(unsigned long long* out, int n) {
    (void)hipDeviceSynchronize();
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(cunvsm::g_dtp_ticks), sizeof(unsigned long long) * static_cast<size_t>(n)));
}
#endif
