// Batch → row order: stable LSD radix sort of (table row, entry id) pairs, hand-written for gfx950.
// Plumbing, not arithmetic: it only permutes entry ids so that update.hip's row passes can own rows; stability is what
// makes every row's gradient sum run in ascending entry order, i.e. run-to-run deterministic.
//
// Keys are table rows: 16-17 bits at the NVSM shape, at most 31. One pass per digit (≤ 9 bits: two passes up to 2^18
// rows, three up to 2^27), two launches per pass, no workspace memsets:
//   * at most 128 workgroups of 8 waves; wave gw owns the contiguous entries [gw·per_wave, (gw+1)·per_wave), so
//     "workgroup, wave, round, lane" order IS ascending entry order;
//   * launch 1 — count: one LDS atomic per entry into the wave's private per-digit counters (order does not matter
//     here), the workgroup publishes its per-digit totals;
//   * launch 2 — scatter: every workgroup recounts its entries (the keys come from L2), turns the published totals into
//     its own scatter bases — digit base (exclusive scan over digits) + the entries of the same digit in earlier
//     workgroups + in earlier waves — and then, per round of 64 entries, the lanes holding the same digit find each other
//     with one ballot per digit bit (no atomics: the rank of an entry among its equals must be its lane order, which makes
//     the sort stable); the lowest lane of each group advances the wave's counter by the group size.
// An earlier form did a pass in ONE launch with a grid-wide meeting point between the two phases. Alone it sorted the
// documents' 870 k pairs in 51 us, but inside a step — on the lowest-priority streams, next to the gather / GEMM / loss
// kernels — a pass took 100-150 us: the meeting point needs all 128 workgroups resident at once, and those that had
// arrived held 8 waves and their LDS each while they waited for the others to be scheduled (a fifth of the loss
// kernel's wave slots). Workgroups of the two-launch form come, work and go.
// (rocPRIM's Onesweep, which this replaces, took 9 launches + 17 workspace memsets per step for the two tables.)
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
constexpr int64_t kSortMinTile = 1024;          // entries per workgroup before a second workgroup is worth its barrier (4096 at first: batch 4096 0.194 -> 0.187 ms per step)

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

// phase 1 of a pass — count: one LDS atomic per entry into the wave's private per-digit counters (order does not matter
// here: a fifth of the instructions of the ballot match the scatter phase cannot do without)
__device__ __forceinline__ void count_digits(const int* __restrict__ keys_in, uint32_t begin, uint32_t end, int shift, uint32_t mask,
                                             int lane, int (&cnt)[kSortMaxBuckets]) {
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
            if (valid) atomicAdd(&cnt[d], 1);
        }
    }
}

// Kernel 1 of a pass: per-workgroup digit totals → counts[block][digit]. (Also clears the caller's per-step counters —
// the CSR's row bounds — instead of a memset launch of their own.)
__global__ __launch_bounds__(kSortThreads) void radix_count_kernel(
        const int* __restrict__ keys_in, uint32_t n, int shift, int D, uint32_t per_wave, int* __restrict__ counts,
        int4* __restrict__ zero_buf, uint32_t zero_n4, const int* __restrict__ gate) {
    __shared__ int cnt[kSortWaves][kSortMaxBuckets];
    if (gate && *gate == 0) return;      // (nothing to sort this time: see sort_pairs)
    for (uint32_t i = blockIdx.x * kSortThreads + threadIdx.x; i < zero_n4; i += gridDim.x * kSortThreads)
        zero_buf[i] = make_int4(0, 0, 0, 0);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int NB = 1 << D;
    const uint32_t mask = static_cast<uint32_t>(NB - 1);
    const uint32_t gw = blockIdx.x * kSortWaves + w;
    const uint64_t b64 = static_cast<uint64_t>(gw) * per_wave;
    const uint32_t begin = b64 < n ? static_cast<uint32_t>(b64) : n;
    const uint32_t end = (b64 + per_wave < n) ? static_cast<uint32_t>(b64 + per_wave) : n;
    for (int i = tid; i < kSortWaves * kSortMaxBuckets; i += kSortThreads) (&cnt[0][0])[i] = 0;
    __syncthreads();
    count_digits(keys_in, begin, end, shift, mask, lane, cnt[w]);
    __syncthreads();
    if (tid < NB) {
        int run = 0;
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ++ww) run += cnt[ww][tid];
        counts[blockIdx.x * kSortMaxBuckets + tid] = run;
    }
}

// Kernel 2 of a pass: recount (the keys come from L2), turn the published totals into this workgroup's scatter bases —
// digit base (exclusive scan over digits) + the entries of the same digit in earlier workgroups + in earlier waves —
// then rank and scatter.
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
        const int* __restrict__ keys_in, const int* __restrict__ vals_in, int* __restrict__ keys_out, int* __restrict__ vals_out,
        uint32_t n, int shift, int D, uint32_t per_wave, const int* __restrict__ counts, const int* __restrict__ gate) {
    __shared__ int cnt[kSortWaves][kSortMaxBuckets];
    __shared__ int wave_tot[kSortWaves];
    if (gate && *gate == 0) return;
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
    count_digits(keys_in, begin, end, shift, mask, lane, cnt[w]);
    __syncthreads();
    // the wave counters become exclusive prefixes over the waves
    if (tid < NB) {
        int run = 0;
#pragma unroll
        for (int ww = 0; ww < kSortWaves; ++ww) { const int c = cnt[ww][tid]; cnt[ww][tid] = run; run += c; }
    }

    // ---- scatter bases ----
    int pre = 0, tot = 0;
    if (tid < NB) {
        const int G = gridDim.x, me = blockIdx.x;
        // sixteen loads in flight per turn, the last turn too (slots past G re-read workgroup G - 1 and count as 0): a
        // one-by-one remainder loop was a chain of up to 15 dependent L2 round trips in a launch that is pure latency
        for (int t = 0; t < G; t += 16) {
            int c[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] = counts[min(t + u, G - 1) * kSortMaxBuckets + tid];
#pragma unroll
            for (int u = 0; u < 16; ++u) { tot += (t + u < G) ? c[u] : 0; pre += (t + u < me) ? c[u] : 0; }
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

    // ---- rank + scatter: per round of 64 entries the lanes holding the same digit find each other with one ballot per
    // digit bit (no atomics: the rank of an entry among its equals must be its lane order, which makes the sort stable) ----
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

// workspace layout: [256 B unused | counts [128][512] int | scratch keys [n] | scratch entries [n]]
size_t sort_pairs_temp_bytes(int64_t n, int /*bits*/) {
    return 256 + sizeof(int) * kSortMaxBlocks * kSortMaxBuckets + 2 * align_up(static_cast<size_t>(n > 0 ? n : 1) * sizeof(int), 256);
}

void sort_pairs(void* temp, size_t temp_bytes, uint64_t* epoch, const int* keys_in, int* keys_out, const int* vals_in,
                int* vals_out, int64_t n, int bits, int* err_flag, hipStream_t s, int* zero_buf, int64_t zero_count, const int* gate) {
    if (n <= 0) return;
    if (zero_count % 4 || reinterpret_cast<uintptr_t>(zero_buf) % 16) throw std::runtime_error("sort_pairs: zero_buf must be 16 B aligned, a multiple of 4 ints");
    if (n >= (int64_t(1) << 31)) throw std::runtime_error("sort_pairs: more than 2^31 entries");
    if (temp_bytes < sort_pairs_temp_bytes(n, bits)) throw std::runtime_error("sort_pairs: workspace too small");
    char* base = static_cast<char*>(temp);
    (void)epoch; (void)err_flag;      // (the one-launch form's arrival counter and time-out report)
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
        hipLaunchKernelGGL(radix_count_kernel, dim3(p.blocks), dim3(kSortThreads), 0, s, src_k, static_cast<uint32_t>(n), shift,
                           p.digit_bits[i], p.per_wave, counts, reinterpret_cast<int4*>(i == 0 ? zero_buf : nullptr),
                           static_cast<uint32_t>(i == 0 && zero_buf ? zero_count / 4 : 0), gate);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(p.blocks), dim3(kSortThreads), 0, s, src_k, src_v, dst_k, dst_v,
                           static_cast<uint32_t>(n), shift, p.digit_bits[i], p.per_wave, counts, gate);
        shift += p.digit_bits[i];
        src_k = dst_k; src_v = dst_v;
    }
}

}  // namespace cunvsm
