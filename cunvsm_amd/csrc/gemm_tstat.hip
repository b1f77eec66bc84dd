// fp32 MFMA GEMM for the two batch-sized projection products of a step, the small operand stationary in LDS:
//   forward   pre[B][d_e]     = phrase[B][d_w] · Tt[d_w][d_e]  (+ bias, + batch-norm column sums)   cpp/params.cu:417
//   backward  gphrase[B][d_w] = alpha · dx[B][d_e] · T          (+ per-row sums of squares)          cpp/objective.cu:453
// M = batch (51 200) is huge, N and K are a few hundred: the projection matrix (307 KB) is the operand every output row
// needs in full. The 128 x 128 tiled kernel (gather_gemm.hip) re-stages it through LDS for every tile behind two barriers
// per 32-deep K step and quantises to 800 tiles on 256 CUs (3.125 tiles per CU = four rounds): 46 % / 36 % of the fp32
// MFMA peak. Here instead
//   * one workgroup per CU (8 waves) keeps a column part of the projection matrix — about half of the N columns, all K
//     rows: 152-160 KB, the whole LDS — resident for its lifetime, in MFMA-fragment order: element (k/4, n) is the float4
//     B[4(k/4) .. 4(k/4)+3][n], so one conflict-free ds_read_b128 feeds four k-steps;
//   * every wave works on its own 16-row blocks of the big operand with NO barrier and NO LDS write in the main loop:
//     lane (i = l & 15, q = l >> 4) loads the float4 A[row0 + i][16 g + 4 q ..+3] straight from global memory (a whole
//     block ahead: ~8 us of latency are covered) and issues v_mfma_f32_16x16x4_f32 with k-slot q of step s standing for
//     k = 16 g + 4 q + s — any fixed pairing of the k's of A and B is a valid dot product;
//   * the operands are fed swapped (the tile is computed transposed), so a lane ends up with four consecutive columns
//     of one output row: 16 B stores, row sums of squares by two cross-lane adds, column sums by a 16-lane DPP reduce;
//   * work is split statically and evenly: column parts get workgroups in proportion to their tile count, a workgroup's
//     row blocks go round-robin to its waves and the blocks left over (nb mod 8) are cut by 16-column tile, so every wave
//     issues the same number of MFMAs to within one tile pass (the same split every run: all sums are deterministic).
// fp32 MFMA is exact f32 (a k-ordered fmaf chain), so results differ from the tiled kernel only by summation order.
#include "kernels.h"
#include "device_utils.h"

#include <atomic>
#include <cstdlib>

namespace cunvsm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTstatWaves = 8;
constexpr int kTstatThreads = kTstatWaves * 64;
constexpr int kTstatMaxParts = 4;
constexpr size_t kTstatLdsBytes = 163840;
constexpr int kTstatMaxDevices = 64;

struct TstatArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    float alpha;
    const float* bias_n;
    double* colstats;        // [2][N] = Σ_rows C, Σ_rows C² (forward: batch-norm statistics) or null
    GridSumWs sums;          // workspace of the ordered sum over the workgroups of a column part (one column group per part)
    float* rowsq;            // [ceil(N/16)][M] = rowsq_scale · Σ_{16 cols of the tile} C² per row, or null
    float rowsq_scale;
    int parts;
    int nt_part[kTstatMaxParts];        // 16-column tiles of each part
    int tile0_part[kTstatMaxParts];     // first tile of each part
    int wg_begin[kTstatMaxParts + 1];   // workgroups [wg_begin[p], wg_begin[p+1]) serve part p
    int nblocks;                        // 16-row blocks = ceil(M / 16)
    float* dump;                        // 16 B nobody reads: where lanes outside the matrix store (see tstat_block)
#ifdef NVSM_TSTAT_DBG
    int dbg;                            // experiments: 1 = no loads of A, 2 = no epilogue, 4 = no LDS reads in the loop
#endif
};
#ifdef NVSM_TSTAT_DBG
#define TSTAT_DBG(g, bit) ((g).dbg & (bit))
#else
#define TSTAT_DBG(g, bit) 0
#endif

// Σ over the 16 lanes of a DPP row (lanes with the same l >> 4); the total lands in lane 15 of the row
__device__ __forceinline__ float row16_sum_to_last(float v) {
    v += dpp_mov<0x111, 0xf>(v);   // row_shr:1
    v += dpp_mov<0x112, 0xf>(v);   // row_shr:2
    v += dpp_mov<0x114, 0xf>(v);   // row_shr:4
    v += dpp_mov<0x118, 0xf>(v);   // row_shr:8
    return v;
}

// groups [G0, G1) of the lane's row of A: float4 A[row][16 g + 4 q ..+3]
template <int KG, int G0, int G1>
__device__ __forceinline__ void tstat_load_a(const TstatArgs& g, int rb, int li, int q, float4 (&a)[KG]) {
    int row = rb * 16 + li;
    row = row < g.M ? row : g.M - 1;                   // rows past the end: harmless re-read, never stored
    const float* p = g.A + static_cast<size_t>(row) * g.lda + 4 * q;
    if (TSTAT_DBG(g, 1)) return;
#pragma unroll
    for (int kg = G0; kg < G1; ++kg) {
        // The last group may reach past K (K % 4 == 0: whole float4s): those lanes re-read the start of their own row —
        // the matching rows of the LDS image are zero — rather than load under a condition (see tstat_block).
        const int off = (kg < KG - 1 || 16 * kg + 4 * q < g.K) ? 16 * kg : 0;
        a[kg] = *reinterpret_cast<const float4*>(p + off);
    }
}

template <int NTC, int NH>
__device__ __forceinline__ void tstat_read_b(const float4* __restrict__ bp, int kg, float4 (&b)[NTC]) {
#pragma unroll
    for (int j = 0; j < NTC; ++j) b[j] = bp[(4 * kg) * NH + 16 * j];
}

// 4 k-steps x NTC tiles: consecutive MFMAs go to different accumulators (v_mfma_f32_16x16x4_f32 issues every 32 cycles
// but a dependent one needs 40)
template <int NTC>
__device__ __forceinline__ void tstat_mma_group(const float4 (&b)[NTC], const float4& a, f32x4 (&acc)[NTC]) {
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].x, a.x, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].y, a.y, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].z, a.z, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].w, a.w, acc[j], 0, 0, 0);
}

// groups [G0, G1) of the K loop; the B fragments of group kg + 1 are read from LDS while group kg is multiplied
template <int KG, int NTC, int NH, int G0, int G1>
__device__ __forceinline__ void tstat_mma_range(const float4* __restrict__ bp, const float4 (&a)[KG], f32x4 (&acc)[NTC]) {
    float4 b0[NTC], b1[NTC];
    tstat_read_b<NTC, NH>(bp, G0, b0);
    // (the scheduling barriers pin "reads of the next group first, then this group's MFMAs": left alone the compiler
    //  sinks the reads to just before their use and every group starts with an exposed LDS round trip)
#pragma unroll
    for (int kg = G0; kg < G1; kg += 2) {
        if (kg + 1 < G1) tstat_read_b<NTC, NH>(bp, kg + 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        tstat_mma_group<NTC>(b0, a[kg], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (kg + 1 < G1) {
            if (kg + 2 < G1) tstat_read_b<NTC, NH>(bp, kg + 2, b0);
            __builtin_amdgcn_sched_barrier(0);
            tstat_mma_group<NTC>(b1, a[kg + 1], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// One 16-row block against tiles [j0, j0 + NTC) of the part: NTC accumulators, KG x 4 x NTC MFMAs. On entry a[0, H) holds
// the block's first K groups; the rest is fetched here behind the first half of the MFMAs, and the first K groups of
// the NEXT block (rb_next; the last task names itself) behind the second half: one register set, loads half a block ahead.
// EPI: 1 = column sums (batch-norm statistics), 2 = row sums of squares, 4 = bias.
// Every global load and store of the loop body is unconditional (lanes outside the matrix load a clamped address and store
// to TstatArgs::dump): the prefetched rows of A are waited for with counted s_waitcnt vmcnt(n), and the compiler can only
// count what is issued on every path — one store under a branch, and it waits for everything issued before it, i.e. for
// the loads of the next half block it has just issued (measured: 79 instead of 66 us for the forward product).
constexpr int kEpiStats = 1, kEpiRowsq = 2, kEpiBias = 4;
template <int KG, int NTC, int NH, int EPI>
__device__ __forceinline__ void tstat_block(const TstatArgs& g, const float4* __restrict__ Bs, int rb, int j0, int tile0,
                                            float4 (&a)[KG], int rb_next, int li, int q, float* __restrict__ stat_slot) {
    constexpr int H = (KG + 1) / 2;
    f32x4 acc[NTC];
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* bp = Bs + q * NH + 16 * j0 + li;
    tstat_load_a<KG, H, KG>(g, rb, li, q, a);
    tstat_mma_range<KG, NTC, NH, 0, H>(bp, a, acc);
    tstat_load_a<KG, 0, H>(g, rb_next, li, q, a);
    tstat_mma_range<KG, NTC, NH, H, KG>(bp, a, acc);
    if (TSTAT_DBG(g, 2)) { if (acc[0][0] == 123.456f) g.C[0] = 1.f; return; }
    // ---- epilogue: acc[j][r] = C[rb 16 + li][16 (tile0 + j0 + j) + 4 q + r] ----
    const int row = rb * 16 + li;
    const bool row_ok = row < g.M;
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
        const int tile = tile0 + j0 + j;
        const int col = 16 * tile + 4 * q;
        const bool col_ok = col < g.N;                   // N % 4 == 0: the lane's four columns are in or out together
        float v[4];
        if (EPI & kEpiBias) {
            const float4 bb = *reinterpret_cast<const float4*>(g.bias_n + (col_ok ? col : 0));
            v[0] = g.alpha * acc[j][0] + bb.x; v[1] = g.alpha * acc[j][1] + bb.y;
            v[2] = g.alpha * acc[j][2] + bb.z; v[3] = g.alpha * acc[j][3] + bb.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = g.alpha * acc[j][r];
        }
        const bool ok = row_ok && col_ok;
        *reinterpret_cast<float4*>(ok ? g.C + static_cast<size_t>(row) * g.ldc + col : g.dump) = make_float4(v[0], v[1], v[2], v[3]);
        if (EPI & kEpiRowsq) {
            // the row's 16 columns of this tile sit in the four lanes li, li + 16, li + 32, li + 48
            float rsq = ok ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
            rsq += __shfl_xor(rsq, 16);
            rsq += __shfl_xor(rsq, 32);
            *((q == 0 && row_ok) ? g.rowsq + static_cast<size_t>(tile) * g.M + row : g.dump) = rsq * g.rowsq_scale;
        }
        if (EPI & kEpiStats) {
            // column sums over the block's 16 rows (= the 16 lanes of a DPP row), added to this wave's private slot
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = ok ? v[r] : 0.f;
                const float s1 = row16_sum_to_last(x);
                const float s2 = row16_sum_to_last(x * x);
                if (li == 15) {
                    // ds_add_f32 without return: no wait for the LDS round trip; the slot is this wave's own and a wave's
                    // LDS operations execute in order, so the sums are the same every run
                    float* sp = stat_slot + 16 * (j0 + j) + 4 * q + r;
                    unsafeAtomicAdd(sp, s1);
                    unsafeAtomicAdd(sp + NH, s2);
                }
            }
        }
    }
}

// KG = ceil(K / 16) groups of 16 k's; NT = tiles of the widest part (LDS row pitch NH = 16 NT; a part has NT or NT - 1
// tiles, MIXED: both occur); BLAY 0: B is [K][N] (ldb), 1: B is stored [N][K] (ldb).
template <int KG, int NT, int BLAY, bool MIXED, int EPI>
__global__ __launch_bounds__(kTstatThreads) void gemm_tstat_kernel(TstatArgs g) {
    extern __shared__ float4 tstat_lds[];
    constexpr int NH = NT * 16;
    constexpr int KQ = KG * 4;                       // float4 rows of the LDS image
    float4* Bs = tstat_lds;
    float* stats = reinterpret_cast<float*>(tstat_lds + KQ * NH);     // [waves][2][NH], only with colstats
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, q = lane >> 4;

    int part = 0;
#pragma unroll
    for (int p = 1; p < kTstatMaxParts; ++p) if (p < g.parts && static_cast<int>(blockIdx.x) >= g.wg_begin[p]) part = p;
    const int nt = g.nt_part[part], tile0 = g.tile0_part[part];
    const int n0 = tile0 * 16;
    const int wgs = g.wg_begin[part + 1] - g.wg_begin[part], me = blockIdx.x - g.wg_begin[part];
    const int base = g.nblocks / wgs, extra = g.nblocks % wgs;
    const int rb_begin = me * base + (me < extra ? me : extra);
    const int nb = base + (me < extra ? 1 : 0);

    // ---- the part's columns of B into LDS, element (kq, n) = B[4 kq .. 4 kq + 3][n0 + n]; zero outside the matrix ----
    if (BLAY == 0) {
        // B[K][N]: a thread takes rows 4 kq .. +3 at columns n0 + 4 n4 .. +3 (four coalesced float4 loads) and transposes
        for (int idx = tid; idx < KQ * (NH / 4); idx += kTstatThreads) {
            const int kq = idx / (NH / 4), n4 = idx - kq * (NH / 4);
            const int col = n0 + 4 * n4;
            float4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 4 * kq + i;
                const bool ok = (k < g.K) && (col < g.N) && (4 * n4 < 16 * nt);
                const float4 v = *reinterpret_cast<const float4*>(g.B + (ok ? static_cast<size_t>(k) * g.ldb + col : 0));
                r[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4* dst = Bs + kq * NH + 4 * n4;
            dst[0] = make_float4(r[0].x, r[1].x, r[2].x, r[3].x);
            dst[1] = make_float4(r[0].y, r[1].y, r[2].y, r[3].y);
            dst[2] = make_float4(r[0].z, r[1].z, r[2].z, r[3].z);
            dst[3] = make_float4(r[0].w, r[1].w, r[2].w, r[3].w);
        }
    } else {
        // B stored [N][K]: element (kq, n) is the float4 at row n0 + n, offset 4 kq; lanes run along kq (coalesced reads)
        for (int idx = tid; idx < KQ * NH; idx += kTstatThreads) {
            const int n = idx / KQ, kq = idx - n * KQ;
            const bool ok = (n0 + n < g.N) && (4 * kq < g.K) && (n < 16 * nt);
            const float4 v = *reinterpret_cast<const float4*>(g.B + (ok ? static_cast<size_t>(n0 + n) * g.ldb + 4 * kq : 0));
            Bs[kq * NH + n] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float* my_stats = nullptr;
    if (EPI & kEpiStats) {
        for (int i = tid; i < kTstatWaves * 2 * NH; i += kTstatThreads) stats[i] = 0.f;
        my_stats = stats + w * 2 * NH;
    }
    __syncthreads();

    // ---- this wave's tasks: full blocks rb_begin + w + 8 t, then a contiguous range of single tiles of the nb % 8 blocks
    // left over (in block-major order, so that consecutive tasks mostly share a block and its rows of A) ----
    constexpr int H = (KG + 1) / 2;
    const int full = nb / kTstatWaves, rem = nb - full * kTstatWaves;
    const int units = rem * nt;
    const int u_begin = (units * w) / kTstatWaves, u_end = (units * (w + 1)) / kTstatWaves;
    const int total = full + (u_end - u_begin);
    auto task_rb = [&](int t) {
        if (t < full) return rb_begin + w + kTstatWaves * t;
        return rb_begin + kTstatWaves * full + (u_begin + (t - full)) / nt;
    };
    float4 a[KG];
    if (total > 0) tstat_load_a<KG, 0, H>(g, task_rb(0), li, q, a);
    // The two waves of a SIMD (w and w + 4) otherwise run the same instruction stream in step — both in their MFMA range,
    // then both in their epilogue and load waits with the matrix pipe idle: the second one starts a few microseconds late
    // (harness, alone: forward 86.8 -> 77.7 us, backward 93.9 -> 85 us; the forward product gains with any offset at all,
    // the backward one up to about five microseconds).
    if (w >= 4) {
#pragma unroll 1
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(32);
    }
    // one loop per block shape (each with a fixed number of loads and stores per iteration, see tstat_block)
    if (!MIXED || nt == NT) {
        for (int t = 0; t < full; ++t)
            tstat_block<KG, NT, NH, EPI>(g, Bs, task_rb(t), 0, tile0, a, task_rb(t + 1 < total ? t + 1 : t), li, q, my_stats);
    } else {
        for (int t = 0; t < full; ++t)
            tstat_block<KG, NT - 1, NH, EPI>(g, Bs, task_rb(t), 0, tile0, a, task_rb(t + 1 < total ? t + 1 : t), li, q, my_stats);
    }
    for (int t = full; t < total; ++t)
        tstat_block<KG, 1, NH, EPI>(g, Bs, task_rb(t), (u_begin + (t - full)) % nt, tile0, a, task_rb(t + 1 < total ? t + 1 : t),
                                    li, q, my_stats);

    if (EPI & kEpiStats) {
        __syncthreads();
        // the eight waves' slots in wave order, then the ordered sum over the part's workgroups (device_utils.h): the same
        // bits every run, where one fp64 atomic per column and workgroup added in arrival order
        int* sum_flag = reinterpret_cast<int*>(tstat_lds);      // (the image of B is dead behind the barrier; LDS is full)
        const GridSumWs& ws = g.sums;
        auto val = [&](int i) -> float {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < kTstatWaves; ++ww) s += stats[ww * 2 * NH + i];
            return s;
        };
        double* cs = g.colstats;
        const int N = g.N;
        auto out = [&](int i, double v) {
            const int st = i / NH, n = i - st * NH;
            if (n < 16 * nt && n0 + n < N) cs[static_cast<size_t>(st) * N + n0 + n] = v;
        };
        grid_sum_ordered<kTstatThreads>(ws.part + static_cast<size_t>(part) * ws.contrib_cap * ws.width_cap,
                                        ws.part2 + static_cast<size_t>(part) * ws.groups_cap * ws.width_cap,
                                        ws.arrive + part * (ws.groups_cap + 1), ws.fan, 2 * NH, me, wgs, val, out, sum_flag);
    }
}

template <int KG, int NT, int BLAY, bool MIXED, int EPI>
static bool tstat_launch_epi(const TstatArgs& g, size_t lds, int wgs, hipStream_t s) {
    // the opt-in to more than 64 KB of dynamic LDS is per DEVICE (a process may hold handles on several GPUs), and several
    // host threads may come through here at once: one flag per device ordinal, set after the call that it stands for
    static std::atomic<bool> attr_set[kTstatMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kTstatMaxDevices) return false;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tstat_kernel<KG, NT, BLAY, MIXED, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kTstatLdsBytes)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev].store(true, std::memory_order_release);
    }
    (void)hipGetLastError();
    NVSM_LAUNCH((gemm_tstat_kernel<KG, NT, BLAY, MIXED, EPI>), dim3(wgs), dim3(kTstatThreads), lds, s, g);
    // a launch the device refuses (LDS opt-in missing, ...) is reported here, so that launch_gemm can fall back to the tiled kernel
    return hipGetLastError() == hipSuccess;
}

// the forward product carries batch-norm column sums or a bias, the backward one row sums of squares or nothing
template <int KG, int NT, int BLAY, bool MIXED>
static bool tstat_launch(const TstatArgs& g, size_t lds, int wgs, hipStream_t s) {
    const int epi = (g.colstats ? kEpiStats : 0) | (g.rowsq ? kEpiRowsq : 0) | (g.bias_n ? kEpiBias : 0);
    if (BLAY == 0) {
        if (epi == kEpiStats) return tstat_launch_epi<KG, NT, BLAY, MIXED, kEpiStats>(g, lds, wgs, s);
        if (epi == kEpiBias) return tstat_launch_epi<KG, NT, BLAY, MIXED, kEpiBias>(g, lds, wgs, s);
        if (epi == 0) return tstat_launch_epi<KG, NT, BLAY, MIXED, 0>(g, lds, wgs, s);
    } else {
        if (epi == kEpiRowsq) return tstat_launch_epi<KG, NT, BLAY, MIXED, kEpiRowsq>(g, lds, wgs, s);
        if (epi == 0) return tstat_launch_epi<KG, NT, BLAY, MIXED, 0>(g, lds, wgs, s);
    }
    return false;
}

#ifdef NVSM_TSTAT_DBG
static int g_tstat_dbg = 0;
#endif

// 256 bytes per device that nobody reads: where lanes outside a matrix store, so that every store of a pipelined loop body
// is unconditional (see tstat_block). Null: allocation failed.
float* gemm_dump_buffer() {
    static std::atomic<float*> dump[kTstatMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kTstatMaxDevices) return nullptr;
    float* p = dump[dev].load(std::memory_order_acquire);
    if (p) return p;
    float* fresh = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&fresh), 256) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    float* expected = nullptr;
    if (!dump[dev].compare_exchange_strong(expected, fresh, std::memory_order_acq_rel)) { (void)hipFree(fresh); return expected; }      // another thread was first
    return fresh;
}
static bool g_gemm_tstat_enabled = true;
void gemm_set_tstat_enabled(bool on) { g_gemm_tstat_enabled = on; }

// returns false when the shape is not one this kernel covers (the caller falls back to the tiled kernel);
// *rowsq_parts = number of [M]-sized parts written to rowsq (one per 16-column tile)
bool launch_gemm_tstat(int a_layout, int b_layout, const float* A, const float* B, float* C, int M, int N, int K,
                       int lda, int ldb, int ldc, float alpha, const float* bias_n, hipStream_t s, double* colstats,
                       float* rowsq, float rowsq_scale, int* rowsq_parts, bool busy_chip, const GridSumWs* sums) {
    // NVSM_GEMM_TSTAT (A/B runs): bit 0 = the forward product (B as [K][N]), bit 1 = the backward one (B stored [N][K])
    const int env_mask = tuning().gemm_tstat;
    if (!g_gemm_tstat_enabled || !(env_mask & (b_layout ? 2 : 1)) || a_layout != 0 || M < 1024) return false;
    // The forward product of a large batch starts while the previous step's documents update and this step's sorts still
    // hold registers and LDS on most CUs, and a kernel that needs a whole CU per workgroup starts late on some of them; its
    // static split has no slack for that. With round 1's step that cost more than the kernel gained (129 us against the
    // tiled kernel's 127 us in-step, step 1.116 against 1.103 ms). Since the documents update's tail and the sorts got
    // shorter it is the other way round at the NVSM shape (1.047 against 1.055 ms, interleaved) — but not where those are
    // long, i.e. with tables much larger than the batch (configs[4]: 1.90 against 1.87 ms): the caller says which
    // (busy_chip). NVSM_GEMM_TSTAT_FWD_ANY=0 / 1 overrides.
    const int fwd_any = tuning().gemm_tstat_fwd_any;
    if (b_layout == 0 && M > 16384 && (fwd_any >= 0 ? fwd_any == 0 : busy_chip)) return false;
    if ((K % 4) || (N % 4) || (lda % 4) || (ldb % 4) || (ldc % 4)) return false;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) % 16) return false;
    if (bias_n && reinterpret_cast<uintptr_t>(bias_n) % 16) return false;
    const int KG = (K + 15) / 16;
    const int tiles = (N + 15) / 16;
    int parts = 0, NT = 0;
    for (int p = 1; p <= kTstatMaxParts; ++p) {
        const int nt = (tiles + p - 1) / p;
        if (nt != 8 && nt != 10) continue;
        const size_t need = static_cast<size_t>(KG) * 4 * nt * 16 * 16 + (colstats ? static_cast<size_t>(kTstatWaves) * 2 * nt * 16 * 4 : 0);
        if (need <= kTstatLdsBytes) { parts = p; NT = nt; break; }
    }
    if (!parts) return false;
    // per device ordinal: the CU count that sizes the grid
    static std::atomic<int> cus_of[kTstatMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kTstatMaxDevices) return false;
    int num_cus = cus_of[dev].load(std::memory_order_acquire);
    if (num_cus == 0) {
        hipDeviceProp_t prop;
        num_cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus_of[dev].store(num_cus, std::memory_order_release);
    }
    float* dump = gemm_dump_buffer();
    if (!dump) return false;
    TstatArgs g{};
    g.dump = dump;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.bias_n = bias_n; g.colstats = colstats; g.rowsq = rowsq; g.rowsq_scale = rowsq_scale;
    g.parts = parts; g.nblocks = (M + 15) / 16;
#ifdef NVSM_TSTAT_DBG
    g.dbg = g_tstat_dbg;
#endif
    // the first `wide` parts have NT tiles, the others NT - 1; workgroups in proportion to the tiles
    const int wide = tiles - parts * (NT - 1);
    int wgs = num_cus < parts ? parts : num_cus;
    int t0 = 0, w0 = 0;
    for (int p = 0; p < parts; ++p) {
        g.nt_part[p] = p < wide ? NT : NT - 1;
        g.tile0_part[p] = t0;
        g.wg_begin[p] = w0;
        t0 += g.nt_part[p];
        const int share = (p + 1 == parts) ? wgs - w0 : static_cast<int>((static_cast<int64_t>(wgs) * t0) / tiles) - w0;
        w0 += share < 1 ? 1 : share;
    }
    g.wg_begin[parts] = w0;
    wgs = w0;
    if (colstats) {
        // the ordered column sums need their workspace, large enough for this launch's split
        if (!sums || sums->colgroups < parts || sums->contrib_cap < wgs || sums->width_cap < 2 * NT * 16) return false;
        g.sums = *sums;
        g.sums.fan = grid_sum_fan(wgs);
        if ((wgs + g.sums.fan - 1) / g.sums.fan > sums->groups_cap) return false;
    }
    const size_t lds = static_cast<size_t>(KG) * 4 * NT * 16 * 16 + (colstats ? static_cast<size_t>(kTstatWaves) * 2 * NT * 16 * 4 : 0);
    bool ok = false;
    const bool mixed = wide < parts;
    if (KG == 19 && NT == 8 && b_layout == 0 && !mixed) ok = tstat_launch<19, 8, 0, false>(g, lds, wgs, s);
    else if (KG == 16 && NT == 10 && b_layout == 1 && mixed) ok = tstat_launch<16, 10, 1, true>(g, lds, wgs, s);
    else if (KG == 8 && NT == 8 && b_layout == 0 && !mixed) ok = tstat_launch<8, 8, 0, false>(g, lds, wgs, s);
    else if (KG == 16 && NT == 8 && b_layout == 1 && !mixed) ok = tstat_launch<16, 8, 1, false>(g, lds, wgs, s);
    if (ok && rowsq_parts) *rowsq_parts = rowsq ? tiles : 0;
    return ok;
}

}  // namespace cunvsm
