// Batch → row order: stable LSD radix sort of (table row, entry id) pairs, hand-written for gfx950.
// Plumbing, not arithmetic: it only permutes entry ids so that update.hip's row passes can own rows; stability is what
// makes every row's gradient sum run in ascending entry order, i.e. run-to-run deterministic.
//
// Keys are table rows: 16-17 bits at the NVSM shape, at most 31. One launch per digit (≤ 9 bits: two launches up to
// 2^18 rows, three up to 2^27), no workspace memsets:
//   * at most 128 workgroups of 8 waves, all co-resident (256 CUs); wave gw owns the contiguous entries
//     [gw·per_wave, (gw+1)·per_wave), so "workgroup, wave, round, lane" order IS ascending entry order;
//   * phase 1 — count: one LDS atomic per entry into the wave's private per-digit counters (order does not matter here);
//   * the workgroup publishes its per-digit totals, all workgroups meet at a grid-wide arrival counter (agent-scope
//     release / acquire; the counter only ever grows — the host passes the value to wait for — so nothing is zeroed
//     between launches), then every workgroup turns the published totals into its own scatter bases: digit base
//     (exclusive scan over digits) + the entries of the same digit in earlier workgroups + in earlier waves;
//   * phase 2 — scatter: per round of 64 entries the lanes holding the same digit find each other with one ballot per digit
//     bit (no atomics: the rank of an entry among its equals must be its lane order, which makes the sort stable); the
//     lowest lane of each group advances the wave's counter by the group size.
// (rocPRIM's Onesweep, which this replaces, took 9 launches + 17 workspace memsets per step for the two tables and was
//  the reason the documents update could not start when the loss kernel finished.)
#include "kernels.h"
#include "device_utils.h"

#include <stdexcept>
#include <string>

namespace cunvsm {

namespace {
constexpr int kSortThreads = 512;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortMaxDigitBits = 9;
constexpr int kSortMaxBuckets = 1 << kSortMaxDigitBits;
constexpr int kSortMaxBlocks = 128;
constexpr int kSortUnroll = 4;
constexpr int64_t kSortMinTile = 4096;          // entries per workgroup before a second workgroup is worth its barrier
constexpr long long kSpinLimit = 1ll << 22;     // ≈ seconds: a barrier that never completes is reported, not hung on

// lanes of this wave (among `active`) whose digit equals mine
__device__ __forceinline__ uint64_t match_digit(uint32_t d, int D, uint64_t active) {
    uint64_t peers = active;
    for (int bit = 0; bit < D; ++bit) {
        const bool set = (d >> bit) & 1u;
        const uint64_t b = __ballot(set);
        peers &= set ? b : ~b;
    }
    return peers;
}

__global__ __launch_bounds__(kSortThreads) void radix_pass_kernel(
        const int* __restrict__ keys_in, const int* __restrict__ vals_in, int* __restrict__ keys_out, int* __restrict__ vals_out,
        uint32_t n, int shift, int D, uint32_t per_wave, int* __restrict__ counts, unsigned long long* sync_counter,
        unsigned long long sync_target, int* __restrict__ err_flag) {
    __shared__ int cnt[kSortWaves][kSortMaxBuckets];
    __shared__ int wave_tot[kSortWaves];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int NB = 1 << D;
    const uint32_t mask = static_cast<uint32_t>(NB - 1);
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t gw = blockIdx.x * kSortWaves + w;
    const uint64_t b64 = static_cast<uint64_t>(gw) * per_wave;
    const uint32_t begin = b64 < n ? static_cast<uint32_t>(b64) : n;
    const uint32_t end = (b64 + per_wave < n) ? static_cast<uint32_t>(b64 + per_wave) : n;

    for (int i = tid; i < kSortWaves * kSortMaxBuckets; i += kSortThreads) (&cnt[0][0])[i] = 0;
    __syncthreads();

    // ---- phase 1: count ----
    for (uint32_t r0 = begin; r0 < end; r0 += 64 * kSortUnroll) {
        int key[kSortUnroll];
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u) {
            const uint32_t i = r0 + u * 64 + lane;
            key[u] = (i < end) ? keys_in[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u) {
            const bool valid = (r0 + u * 64 + lane) < end;
            const uint32_t d = (static_cast<uint32_t>(key[u]) >> shift) & mask;
            // counting needs no order: an LDS atomic per entry (a fifth of the instructions of the ballot match below, which
            // the scatter phase cannot do without — the sorts share the chip with the step's bandwidth-bound kernels)
            if (valid) atomicAdd(&cnt[w][d], 1);
        }
    }
    __syncthreads();
    // per-digit totals of this workgroup; the wave counters become exclusive prefixes over the waves
    if (tid < NB) {
        int run = 0;
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ++ww) { const int c = cnt[ww][tid]; cnt[ww][tid] = run; run += c; }
        // published write-through (agent-scope atomic store = sc1): readable by the other XCDs without any cache-wide
        // write-back / invalidate
        __hip_atomic_store(&counts[blockIdx.x * kSortMaxBuckets + tid], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- grid-wide meeting point (cdna_hip_programming.md §6 G16, the write-through-payload form) ----
    // The payload travels as agent-scope atomics (stores above, loads below), drained before the arrival counter is
    // bumped; the spin is a relaxed load. No acquire / release FENCE anywhere: an agent-scope acquire per spin iteration
    // invalidates the CU's vector cache and the XCD's L2 lines each time — measured: the word gather-mean running next
    // to the sorts went from 67 to 153 us and the projection GEMM from 106 to 161 us.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(sync_counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        while (__hip_atomic_load(sync_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sync_target) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > kSpinLimit) { *err_flag = NVSM_SORT_TIMEOUT; break; }
        }
    }
    __syncthreads();

    // ---- scatter bases ----
    int pre = 0, tot = 0;
    if (tid < NB) {
        const int G = gridDim.x, me = blockIdx.x;
        int t = 0;
        for (; t + 16 <= G; t += 16) {
            int c[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] = __hip_atomic_load(&counts[(t + u) * kSortMaxBuckets + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 16; ++u) { tot += c[u]; pre += (t + u < me) ? c[u] : 0; }
        }
        for (; t < G; ++t) {
            const int c = __hip_atomic_load(&counts[t * kSortMaxBuckets + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot += c; pre += (t < me) ? c : 0;
        }
    }
    // exclusive scan of tot over the digits (tid): wave-inclusive by shuffles, then the wave totals
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int wave_off = 0;
#pragma unroll
    for (int ww = 0; ww < kSortWaves; ++ww) wave_off += (ww < w) ? wave_tot[ww] : 0;
    if (tid < NB) {
        const int base = wave_off + incl - tot + pre;
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ++ww) cnt[ww][tid] += base;
    }
    __syncthreads();

    // ---- phase 2: rank + scatter ----
    for (uint32_t r0 = begin; r0 < end; r0 += 64 * kSortUnroll) {
        int key[kSortUnroll], val[kSortUnroll];
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u) {
            const uint32_t i = r0 + u * 64 + lane;
            const bool valid = i < end;
            key[u] = valid ? keys_in[i] : 0;
            val[u] = vals_in ? (valid ? vals_in[i] : 0) : static_cast<int>(i);
        }
#pragma unroll
        for (int u = 0; u < kSortUnroll; ++u) {
            if (r0 + u * 64 >= end) break;
            const bool valid = (r0 + u * 64 + lane) < end;
            const uint32_t d = (static_cast<uint32_t>(key[u]) >> shift) & mask;
            const uint64_t peers = match_digit(d, D, __ballot(valid));
            const int pos = cnt[w][d] + __popcll(peers & lt);
            if (valid && (peers & lt) == 0) cnt[w][d] += __popcll(peers);     // after every lane of the group has read it
            if (valid) { keys_out[pos] = key[u]; vals_out[pos] = val[u]; }
        }
    }
}

struct SortPlan { int passes; int digit_bits[4]; int blocks; uint32_t per_wave; };

SortPlan make_plan(int64_t n, int bits) {
    SortPlan p{};
    if (bits < 1) bits = 1;
    p.passes = (bits + kSortMaxDigitBits - 1) / kSortMaxDigitBits;
    int left = bits;
    for (int i = 0; i < p.passes; ++i) {                      // as even as possible, low digit first: 17 → 9 + 8
        const int d = (left + (p.passes - i) - 1) / (p.passes - i);
        p.digit_bits[i] = d;
        left -= d;
    }
    int64_t g = (n + kSortMinTile - 1) / kSortMinTile;
    if (g < 1) g = 1;
    if (g > kSortMaxBlocks) g = kSortMaxBlocks;
    p.blocks = static_cast<int>(g);
    const int64_t waves = g * kSortWaves;
    int64_t pw = (n + waves - 1) / waves;
    pw = (pw + 63) / 64 * 64;
    p.per_wave = static_cast<uint32_t>(pw);
    return p;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
}  // namespace

// workspace layout: [arrival counter (u64, padded to 256 B) | counts [128][512] int | scratch keys [n] | scratch entries [n]]
// The counter must be zero when the workspace is first used (the caller zero-fills the allocation once) and only grows.
size_t sort_pairs_temp_bytes(int64_t n, int /*bits*/) {
    return 256 + sizeof(int) * kSortMaxBlocks * kSortMaxBuckets + 2 * align_up(static_cast<size_t>(n > 0 ? n : 1) * sizeof(int), 256);
}

void sort_pairs(void* temp, size_t temp_bytes, uint64_t* epoch, const int* keys_in, int* keys_out, const int* vals_in,
                int* vals_out, int64_t n, int bits, int* err_flag, hipStream_t s) {
    if (n <= 0) return;
    if (n >= (int64_t(1) << 31)) throw std::runtime_error("sort_pairs: more than 2^31 entries");
    if (temp_bytes < sort_pairs_temp_bytes(n, bits)) throw std::runtime_error("sort_pairs: workspace too small");
    char* base = static_cast<char*>(temp);
    unsigned long long* counter = reinterpret_cast<unsigned long long*>(base);
    int* counts = reinterpret_cast<int*>(base + 256);
    const size_t arr = align_up(static_cast<size_t>(n) * sizeof(int), 256);
    int* tmp_k = reinterpret_cast<int*>(base + 256 + sizeof(int) * kSortMaxBlocks * kSortMaxBuckets);
    int* tmp_v = reinterpret_cast<int*>(reinterpret_cast<char*>(tmp_k) + arr);
    const SortPlan p = make_plan(n, bits);
    // ping-pong so that the last pass lands in the output arrays
    const int* src_k = keys_in; const int* src_v = vals_in;
    int shift = 0;
    for (int i = 0; i < p.passes; ++i) {
        const bool to_out = ((p.passes - 1 - i) % 2) == 0;
        int* dst_k = to_out ? keys_out : tmp_k;
        int* dst_v = to_out ? vals_out : tmp_v;
        *epoch += static_cast<uint64_t>(p.blocks);
        hipLaunchKernelGGL(radix_pass_kernel, dim3(p.blocks), dim3(kSortThreads), 0, s, src_k, src_v, dst_k, dst_v,
                           static_cast<uint32_t>(n), shift, p.digit_bits[i], p.per_wave, counts, counter,
                           static_cast<unsigned long long>(*epoch), err_flag);
        shift += p.digit_bits[i];
        src_k = dst_k; src_v = dst_v;
    }
}

}  // namespace cunvsm
