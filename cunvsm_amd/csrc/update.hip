// gfx950 kernels: batch → CSR by table row, segmented row passes (the sparse optimisers), dense
// projection optimiser.
//
// The reference applies sparse gradients with one fp32 atomicAdd per (entry, element)
// (update_repr_kernel, cpp/storage.cu:37-49: 614 MB + 891 MB of atomics per step at the NVSM config,
// hot words serialised) and needs separate dense passes for the L2 / Adam decay. Here the batch's
// (row, entry) pairs are radix-sorted once per table and every table row is owned by one thread
// group, which streams the gradient rows of its entries with coalesced 16 B loads, applies the decay
// and the optimiser's row-local formula, and writes the row once: no atomics, deterministic sums,
// decay and update fused in a single pass over the table.
#include "kernels.h"
#include "device_utils.h"

#include <cstdlib>

namespace cunvsm {

// =============================================================================================
// CSR construction from the sorted (row key, entry) pairs
// =============================================================================================
// (round 5: the chunk ranges of the long rows are reserved here too. A row is long iff the entry a chunk's length in front of its
//  LAST entry carries the same key — one more load for the lane that closes the row, which then finds the row's first entry by
//  bisection (rows that long are a few hundred hot words; no documents row at |D| = 2 M) — instead of a launch of its own that read
//  both bounds of EVERY table row: 141 us on the documents CSR chain at |D| = 2 M, 5-10 us and a launch per table elsewhere.)
// (round 6: ... and, chunk_desc given, the wave that reserved a long row's chunks writes their descriptors too — it knows the row's
//  first and last entry and the reserved ranges, which is all csr_chunk_fill_kernel looks up per entry: bounds + reservation + fill
//  are ONE launch of a small batch's CSR build, a chain of launch latencies next to the forward pass.)
__global__ void csr_bounds_kernel(const int* __restrict__ key, int64_t n, int* __restrict__ row_begin,
                                  int* __restrict__ row_end, int* __restrict__ touched, int* __restrict__ num_touched,
                                  int* __restrict__ chunk_base, int* __restrict__ chunk2_base, int* __restrict__ num_chunks, int chunk,
                                  int* __restrict__ chunk_desc, int* __restrict__ chunk2_desc, int max_chunks, int max_chunks2) {
    const int lane = threadIdx.x & 63;
    const uint64_t lt = (1ull << lane) - 1ull;
    // (whole waves iterate together: the touched-row list is appended to with ONE atomic per wave and turn — one per row
    //  start serialised 700 k atomics on a single counter at |D| = 2 M: 158 us)
    for (int64_t i0 = blockIdx.x * static_cast<int64_t>(blockDim.x) + (threadIdx.x - lane); i0 < n;
         i0 += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t i = i0 + lane;
        const bool in = i < n;
        const int k = in ? key[i] : 0;
        const bool head = in && (i == 0 || key[i - 1] != k);
        if (head) row_begin[k] = static_cast<int>(i);
        const bool tail = in && (i == n - 1 || key[i + 1] != k);
        if (tail) row_end[k] = static_cast<int>(i + 1);
        // rows of more than kChunk entries that end in this wave's range, one after the other: the whole wave looks for the row's
        // first entry — 64 probes per round trip, three rounds for 2^18 entries (a lane bisecting on its own was twenty dependent
        // loads, each of them microseconds next to the loss kernel: the words CSR came 50 us late at batch 51 200)
        uint64_t long_tails = __ballot(chunk_base && tail && i >= chunk && key[i - chunk] == k);
        while (long_tails) {
            const int l = __ffsll(static_cast<long long>(long_tails)) - 1;
            long_tails &= long_tails - 1;
            const int kk = __builtin_amdgcn_readlane(k, l);
            const int last = __builtin_amdgcn_readlane(static_cast<int>(i), l);      // (n < 2^31: sort_pairs)
            int lo = 0, hi = last - chunk;                               // the first position of key kk lies in [lo, hi]
            while (lo < hi) {
                const int64_t span = hi - lo;
                const int p = lo + static_cast<int>(span * lane / 64);    // lane 0 probes lo
                const int less = __popcll(__ballot(key[p] < kk));       // the keys are sorted: lanes 0 .. less - 1
                const int nlo = less == 0 ? lo : lo + static_cast<int>(span * (less - 1) / 64) + 1;
                const int nhi = less == 64 ? hi : lo + static_cast<int>(span * less / 64);
                lo = nlo; hi = nhi;
            }
            const int nch = (last + 1 - lo + chunk - 1) / chunk;         // (lo is the same in every lane)
            int base = 0, base2 = 0;
            if (lane == l) {
                base = atomicAdd(num_chunks, nch);
                chunk_base[kk] = base;
                if (nch > kFan) { base2 = atomicAdd(num_chunks + 1, (nch + kFan - 1) / kFan); chunk2_base[kk] = base2; }
            }
            if (chunk_desc) {                                            // what csr_chunk_fill_kernel writes for this row
                base = __shfl(base, l, 64); base2 = __shfl(base2, l, 64);
                for (int c = lane; c < nch; c += 64) {
                    if (base + c < max_chunks) {
                        const int first = lo + c * chunk;
                        chunk_desc[(base + c) * 3 + 0] = kk;
                        chunk_desc[(base + c) * 3 + 1] = first;
                        chunk_desc[(base + c) * 3 + 2] = min(last + 1, first + chunk);
                    }
                }
                if (nch > kFan) {
                    const int nch2 = (nch + kFan - 1) / kFan;
                    for (int c2 = lane; c2 < nch2; c2 += 64) {
                        if (base2 + c2 < max_chunks2) {
                            chunk2_desc[(base2 + c2) * 2 + 0] = base + c2 * kFan;
                            chunk2_desc[(base2 + c2) * 2 + 1] = base + min(nch, (c2 + 1) * kFan);
                        }
                    }
                }
            }
        }
        if (touched) {                                                   // list order is irrelevant: rows are independent
            const uint64_t heads = __ballot(head);
            if (heads) {
                const int leader = __ffsll(static_cast<long long>(heads)) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(num_touched, __popcll(heads));
                base = __shfl(base, leader, 64);
                if (head) touched[base + __popcll(heads & lt)] = k;
            }
        }
    }
}

// Rows with more than kChunk entries (hot words) are cut into chunks of kChunk entries that are reduced in
// parallel (level 1); rows with more than kFan level-1 chunks additionally get level-2 chunks, each the ordered
// sum of kFan level-1 partials, so that no thread group ever walks more than max(kChunk, kFan) items serially
// (until a row exceeds kChunk·kFan² entries) and every sum has a fixed order: results are run-to-run deterministic.
// csr_chunks_kernel (large batches) / csr_bounds_kernel itself (small ones: launch_csr_build) reserve the chunk ranges of the long
// rows; the descriptors are written by
// csr_chunk_fill_kernel, one thread per entry (a row-serial fill took 94 us for the Zipf head word).
__global__ void csr_chunks_kernel(const int* __restrict__ row_begin, const int* __restrict__ row_end, int64_t rows,
                                  int* __restrict__ chunk_base, int* __restrict__ chunk2_base, int* __restrict__ num_chunks, int chunk) {
    for (int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < rows;
         r += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int cnt = row_end[r] - row_begin[r];
        if (cnt > chunk) {
            const int nch = (cnt + chunk - 1) / chunk;
            chunk_base[r] = atomicAdd(num_chunks, nch);
            if (nch > kFan) chunk2_base[r] = atomicAdd(num_chunks + 1, (nch + kFan - 1) / kFan);
        }
    }
}

__global__ void csr_chunk_fill_kernel(const int* __restrict__ key, int64_t n, const int* __restrict__ row_begin,
                                      const int* __restrict__ row_end, const int* __restrict__ chunk_base,
                                      const int* __restrict__ chunk2_base, int* __restrict__ chunk_desc,
                                      int* __restrict__ chunk2_desc, int max_chunks, int max_chunks2, const int* __restrict__ num_chunks,
                                      const int* __restrict__ sorted_entry, int* __restrict__ order_key, int chunk) {
    // no row of more than kChunk entries this step (the documents table, most steps): nothing to describe — the kernel would
    // otherwise read three words per entry to find that out (27 us next to the loss kernel for 870 k entries)
    if (num_chunks[0] == 0) return;
    // the key of the chunks' batch order (launch_chunk_order) is written on the way: the entry that opens a chunk knows the
    // chunk's first entry; slots behind the chunks in use sort last (one launch less per build)
    if (order_key) {
        const int nchunks = min(num_chunks[0], max_chunks);
        for (int ci = nchunks + blockIdx.x * blockDim.x + threadIdx.x; ci < max_chunks; ci += gridDim.x * blockDim.x) order_key[ci] = 511;
    }
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int r = key[i];
        const int b = row_begin[r], e = row_end[r];
        const int cnt = e - b;
        const int rel = static_cast<int>(i) - b;
        if (cnt <= chunk || rel % chunk != 0) continue;
        const int c = rel / chunk;                        // this entry opens level-1 chunk c of row r
        const int nch = (cnt + chunk - 1) / chunk;
        const int base = chunk_base[r];
        if (base + c < max_chunks) {
            chunk_desc[(base + c) * 3 + 0] = r;
            chunk_desc[(base + c) * 3 + 1] = static_cast<int>(i);
            chunk_desc[(base + c) * 3 + 2] = min(e, static_cast<int>(i) + chunk);
            if (order_key) {
                const uint64_t first = static_cast<uint32_t>(sorted_entry[i]);
                order_key[base + c] = static_cast<int>(min<uint64_t>(511, first * 512 / static_cast<uint64_t>(n)));
            }
        }
        if (nch > kFan && c % kFan == 0) {                // ... and level-2 chunk c / kFan
            const int c2 = c / kFan;
            const int base2 = chunk2_base[r];
            if (base2 + c2 < max_chunks2) {
                chunk2_desc[(base2 + c2) * 2 + 0] = base + c;
                chunk2_desc[(base2 + c2) * 2 + 1] = base + min(nch, c + kFan);
            }
        }
    }
}

// Workgroups per CSR kernel (they grid-stride). For a table much larger than the batch the bounds / chunk kernels are
// scattered 4-byte accesses into row-indexed arrays of many MB; spread over every CU they sit next to the loss kernel and
// slow its row gathers (378 vs 217 us at |D| = 2 M) — a few waves per CU on half the CUs do the same work in the time they
// have (configs[4]: 1.99 -> 1.93 ms per step; no effect at the NVSM shape, where the arrays live in L2). Round 4: 64 instead of
// 128 — these workgroups live as long as their kernel (150 us next to the loss kernel at |D| = 2 M), and every CU one of them
// sits on is closed to the backward projection product behind the loss kernel, whose workgroups need a CU's whole register file
// (173 us there for 65 alone): |V| = 500 k, |D| = 2 M 1.717 -> 1.668 ms (96: 1.675, 48: 1.647, 32: 1.61-1.65, 16: 1.79, 8: 2.19,
// uncapped: 1.71), batch 6 400 0.298 -> 0.292, LSE 0.168 -> 0.164. NVSM_CSR_GRID_CAP overrides for every table (experiments).
constexpr int64_t kCsrMergeMaxEntries = 64 * 4096;      // batches below it: the bounds kernel reserves the long rows' chunk ranges (also the default of the entry walk's threshold, tuning.h entry_walk_min; a table gets a chunk ORDER from 60 * 4096 entries: model.cpp alloc_table)
static int csr_grid(int64_t items, bool sparse_table) {
    const int cap_env = tuning().csr_grid_cap;
    const int cap = cap_env >= 0 ? cap_env : (sparse_table ? 64 : 0);
    const int g = stream_grid(items, 256);
    return (cap > 0 && g > cap) ? cap : g;
}

void launch_csr_build(const Csr& c, hipStream_t s, bool counters_cleared, int* order_key) {
    // row_begin | row_end | num_chunks | num_touched are one allocation (model.cpp), padded so that one fill kernel does it
    const bool sparse = row_pass_split(c);
    if (!counters_cleared) (void)hipMemsetAsync(c.row_begin, 0, sizeof(int) * csr_counter_ints(c.rows), s);
    // Small batches (a CSR build there is a chain of launch latencies, and a launch the host has to queue: the LSE step is
    // within 5 % of host-bound): the bounds kernel reserves the long rows' chunks itself — LSE 0.1495 -> 0.1477 ms, batch 6 400
    // 0.2413 -> 0.2389. Large batches keep the launch of their own: the build is off their critical path, and WITHOUT it the
    // updates it feeds start earlier and take bandwidth from the main stream (batch 51 200 +0.7 %, |D| = 2 M +0.6 %).
    const bool reserve_in_bounds = c.n > 0 && c.n < kCsrMergeMaxEntries;
    // ... and can write their descriptors too (round 6, NVSM_CSR_FILL_IN_BOUNDS=1: one launch less again, bit-identical — but the wave
    // that closes a hot row then serialises bisection + descriptor writes inside the bounds kernel, and the LSE step, whose CSR chain is
    // co-critical, got 3 % SLOWER (batch 6 400 / 3 200 unchanged): off by default, NOTES_r06 §3), unless the chunks get a batch order
    // (handles made for large batches: the fill kernel writes the order's keys, whose unused slots need the final chunk count)
    const bool fill_in_bounds = reserve_in_bounds && !(c.chunk_order && order_key) && tuning().csr_fill_in_bounds;
    if (c.n > 0)
        hipLaunchKernelGGL(csr_bounds_kernel, dim3(csr_grid(c.n, sparse)), dim3(256), 0, s, c.sorted_key, c.n, c.row_begin, c.row_end,
                           row_pass_split(c) ? c.touched : nullptr, c.num_touched, reserve_in_bounds ? c.chunk_base : nullptr, c.chunk2_base, c.num_chunks, c.chunk,
                           fill_in_bounds ? c.chunk_desc : nullptr, c.chunk2_desc, c.max_chunks, c.max_chunks2);
    if (!reserve_in_bounds)
        hipLaunchKernelGGL(csr_chunks_kernel, dim3(csr_grid(c.rows, sparse)), dim3(256), 0, s, c.row_begin, c.row_end, c.rows,
                           c.chunk_base, c.chunk2_base, c.num_chunks, c.chunk);
    if (c.n > 0 && !fill_in_bounds)
        hipLaunchKernelGGL(csr_chunk_fill_kernel, dim3(csr_grid(c.n, sparse)), dim3(256), 0, s, c.sorted_key, c.n, c.row_begin,
                           c.row_end, c.chunk_base, c.chunk2_base, c.chunk_desc, c.chunk2_desc, c.max_chunks, c.max_chunks2, c.num_chunks,
                           c.sorted_entry, c.chunk_order ? order_key : nullptr, c.chunk);
}

// =============================================================================================
// Segment accumulation: g = Σ coef·X[src][col..col+V), q = Σ sq, over sorted entries [begin, end), in entry order.
// kSegUnroll gradient rows are in flight per lane, and the entry ids of the NEXT batch are fetched while the
// current batch's rows load, so the dependent chain per batch is one global-load latency (ids → rows would be
// two). Out-of-range slots of the last batch re-read the segment's last entry with coefficient 0 (branch-free).
// src = entry / div by multiply-shift: exact for entry < 2^26, div <= 2048 (checked by the host).
// =============================================================================================
// Source rows in flight per lane in the gather of a row's / chunk's entries. Eight (one round for a row of up to eight
// entries) was the first choice; interleaved A/B over 3-8 for either table at the NVSM shape says five for both: 1.030 ->
// 1.007 ms per step. Not a per-kernel optimum — alone the passes like eight — but the two updates run next to each other
// and share the memory system: w8 d5 1.062, w5 d8 1.028, w5 d5 1.010, w5 d4 1.055 (97 registers: a fifth wave per SIMD for
// the documents pass, which then crowds out the words chain the step waits for), w5 d6 1.022, w4 d5 1.011, w6 d5 1.011.
#ifndef NVSM_SEG_UNROLL_WORDS
#define NVSM_SEG_UNROLL_WORDS 5
#endif
#ifndef NVSM_SEG_UNROLL_DOCS
#define NVSM_SEG_UNROLL_DOCS 5
#endif
template <int TABLE> struct SegUnrollDeep { static constexpr int value = TABLE == 0 ? NVSM_SEG_UNROLL_WORDS : NVSM_SEG_UNROLL_DOCS; };
// ... in the WIDE family of the one-launch table pass (the Adam modes: five waves per SIMD, table_pass_wide_kernel): with a
// quarter more rows in flight per CU the best depth is lower — w4 d3 against w5 d5: batch 51 200 0.879 -> 0.860 ms, 6 400
// 0.251 -> 0.249, 25 600 0.563 -> 0.5636, full_adam 0.762 -> 0.757 (w4 d4 / w3 d3 / w2 d4 within 0.5 % of it; six spills)
#ifndef NVSM_SEG_UNROLL_WIDE_WORDS
#define NVSM_SEG_UNROLL_WIDE_WORDS 4
#endif
#ifndef NVSM_SEG_UNROLL_WIDE_DOCS
#define NVSM_SEG_UNROLL_WIDE_DOCS 3
#endif
template <int TABLE> struct SegUnrollWide { static constexpr int value = TABLE == 0 ? NVSM_SEG_UNROLL_WIDE_WORDS : NVSM_SEG_UNROLL_WIDE_DOCS; };
// rows of a table much larger than the batch hold one or two entries: two slots in flight per lane leave registers for
// 2-3x as many rows in flight per CU, which is what bounds that regime (a dependent chain of four loads per row)
// ROW_SCALAR_ACC (the accumulator pass of Adagrad's words update) gathers scalars only. Sixteen entries in flight at first ("a handful
// of registers") — but the kernel is compiled for it at 114 registers, four waves per SIMD, where four in flight give 74 and six
// waves, and the pass is a few thousand short rows bounded by how many are in flight: LSE batch 4 096 0.1480 -> 0.1412 ms (8: 0.1426,
// 6: 0.142, 3: 0.137 / 2: 0.1385 against 0.135 on another box), Adagrad at batch 51 200 0.751 -> 0.726
constexpr int kSegUnrollScalar = 4;
constexpr int kSegUnrollShallow = 3;      // (2 at first; round 2: 4 — LSE batch 4096 0.197 -> 0.193 ms —; round 5, after the families: 3 — LSE 0.1493 -> 0.148, 0.1429 -> 0.1406 on another box, |D| = 2 M -0.6 %, batch 6 400 / 9 600 unchanged; 5 / 6: LSE +1 / +4 %)

template <int V, int TABLE, bool VEC, int kSegUnroll = SegUnrollDeep<TABLE>::value>
__device__ __forceinline__ void accumulate_segment(const RowPassArgs& a, const int* __restrict__ sorted_entry,
                                                   int begin, int end, int col, float (&g)[V], float& q) {
    // Which products the compiler fuses with the following add may differ from one kernel this is inlined into to the
    // next; the three-launch and the one-launch form of a pass (and the dense / streaming halves of a split pass) must
    // round alike, so the contraction is spelled out: fused here, never in apply_row_formula.
#pragma clang fp contract(off)
    const bool need_q = (a.sq_src != nullptr);
    const int last = end - 1;
    int en_next[kSegUnroll];
#pragma unroll
    for (int u = 0; u < kSegUnroll; ++u) en_next[u] = sorted_entry[min(begin + u, last)];
    for (int e = begin; e < end; e += kSegUnroll) {
        uint32_t en[kSegUnroll];
#pragma unroll
        for (int u = 0; u < kSegUnroll; ++u) en[u] = static_cast<uint32_t>(en_next[u]);
        if (e + kSegUnroll < end) {
#pragma unroll
            for (int u = 0; u < kSegUnroll; ++u) en_next[u] = sorted_entry[min(e + kSegUnroll + u, last)];
        }
        float cf[kSegUnroll], sq[kSegUnroll], x[kSegUnroll][V];
#pragma unroll
        for (int u = 0; u < kSegUnroll; ++u) {
            const uint32_t src = static_cast<uint32_t>((static_cast<uint64_t>(en[u]) * a.div_magic) >> 37);
            float c;
            if (TABLE == 0) c = a.wts ? a.wts[en[u]] : 1.f;
            else c = a.coefs ? a.coefs[en[u]] : 1.f;
            sq[u] = need_q ? a.sq_src[src] : 0.f;
            const float scl = (TABLE == 0 && a.src_scale) ? a.src_scale[src] : 1.f;
            const bool ok = (e + u) < end;
            c = ok ? c : 0.f;
            // q uses the unscaled coefficient (cpp/updates_adagrad.cu:136-158 / updates_adam.cu:232-240)
            sq[u] = (TABLE == 0) ? c * sq[u] : (c * c) * sq[u];
            cf[u] = (TABLE == 0 && a.src_scale) ? c * scl : c;
            if (VEC) ldv<V>(a.X + static_cast<size_t>(src) * a.dim + col, x[u]);
        }
#pragma unroll
        for (int u = 0; u < kSegUnroll; ++u) {
            if (need_q) q += sq[u];
            if (VEC) {
#pragma unroll
                for (int i = 0; i < V; ++i) g[i] = __builtin_fmaf(cf[u], x[u][i], g[i]);
            }
        }
    }
}

// ordered sum of `count` partial vectors starting at partial[first]: U loads in flight, added in index order (the sum is
// the same sequence of additions whatever U is). U = 16 for the level-2 pass, whose few thread groups (a dozen very hot
// rows) each walk 64 partials and are pure latency: 16 dependent round trips of 4 loads took 19-25 us in the step.
template <int V, bool VEC, int U>
__device__ __forceinline__ void sum_partials(const float* __restrict__ partial, const float* __restrict__ partial_q,
                                             int first, int count, int dim, int col, float (&g)[V], float& q) {
    int ch = 0;
    for (; ch + U <= count; ch += U) {
        float x[U][V], pq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (VEC) ldv<V>(partial + static_cast<size_t>(first + ch + u) * dim + col, x[u]);
            pq[u] = partial_q[first + ch + u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (VEC) {
#pragma unroll
                for (int i = 0; i < V; ++i) g[i] += x[u][i];
            }
            q += pq[u];
        }
    }
    for (; ch < count; ++ch) {
        if (VEC) {
            float x[V];
            ldv<V>(partial + static_cast<size_t>(first + ch) * dim + col, x);
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] += x[i];
        }
        q += partial_q[first + ch];
    }
}


// ---- hand-over of partial sums between thread groups of ONE launch (table_pass_kernel) -----------------------------------
// The XCDs' L2s are not coherent with each other inside a kernel. A partial that another workgroup will read travels as
// agent-scope atomic stores / loads (write-through, cache-bypassing: the form the radix sort's grid-wide meeting point
// uses, sort.hip) — no release / acquire fence, which would write back and invalidate whole caches under the step's
// bandwidth-bound kernels.
template <int V>
__device__ __forceinline__ void st_agent(float* p, const float (&x)[V]) {
    if constexpr (V == 4) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
        const unsigned long long lo = static_cast<unsigned long long>(__float_as_uint(x[0])) | (static_cast<unsigned long long>(__float_as_uint(x[1])) << 32);
        const unsigned long long hi = static_cast<unsigned long long>(__float_as_uint(x[2])) | (static_cast<unsigned long long>(__float_as_uint(x[3])) << 32);
        __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i)
            __hip_atomic_store(reinterpret_cast<unsigned int*>(p) + i, __float_as_uint(x[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int V>
__device__ __forceinline__ void ld_agent(const float* p, float (&x)[V]) {
    if constexpr (V == 4) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<float*>(p));
        const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x[0] = __uint_as_float(static_cast<unsigned int>(lo)); x[1] = __uint_as_float(static_cast<unsigned int>(lo >> 32));
        x[2] = __uint_as_float(static_cast<unsigned int>(hi)); x[3] = __uint_as_float(static_cast<unsigned int>(hi >> 32));
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i)
            x[i] = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned int*>(const_cast<float*>(p)) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
}

// sum_partials over partials written by other workgroups of the same launch: the same additions in the same order
template <int V, bool VEC, int U>
__device__ __forceinline__ void sum_partials_agent(const float* partial, const float* partial_q, int first, int count, int dim,
                                                   int col, float (&g)[V], float& q) {
    int ch = 0;
    for (; ch + U <= count; ch += U) {
        float x[U][V], pq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (VEC) ld_agent<V>(partial + static_cast<size_t>(first + ch + u) * dim + col, x[u]);
            pq[u] = ld_agent1(partial_q + first + ch + u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (VEC) {
#pragma unroll
                for (int i = 0; i < V; ++i) g[i] += x[u][i];
            }
            q += pq[u];
        }
    }
    for (; ch < count; ++ch) {
        if (VEC) {
            float x[V];
            ld_agent<V>(partial + static_cast<size_t>(first + ch) * dim + col, x);
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] += x[i];
        }
        q += ld_agent1(partial_q + first + ch);
    }
}

// level 1: one thread group (nvec threads, one 16 B column each) per chunk of a long row
template <int V, int TABLE, bool VEC>
__global__ __launch_bounds__(256) void chunk_pass_kernel(Csr c, RowPassArgs a, int G, int nvec) {
    const int gpb = blockDim.x / G;
    const int group = threadIdx.x / G, lig = threadIdx.x - group * G;
    if (group >= gpb) return;
    const int nchunks = min(c.num_chunks[0], c.max_chunks);
    for (int ci = blockIdx.x * gpb + group; ci < nchunks; ci += gridDim.x * gpb) {
        const int begin = c.chunk_desc[ci * 3 + 1], end = c.chunk_desc[ci * 3 + 2];
        for (int cv = lig; cv < nvec; cv += G) {
            const int col = cv * V;
            float g[V];
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] = 0.f;
            float q = 0.f;
            accumulate_segment<V, TABLE, VEC, VEC ? SegUnrollDeep<TABLE>::value : kSegUnrollScalar>(a, c.sorted_entry, begin, end, col, g, q);
            if (VEC) stv<V>(c.partial + static_cast<size_t>(ci) * a.dim + col, g);
            if (cv == 0) c.partial_q[ci] = q;
        }
    }
}

// level 2: ordered sum of up to kFan level-1 partials
template <int V, bool VEC>
__global__ __launch_bounds__(256) void chunk2_pass_kernel(Csr c, int dim, int G, int nvec) {
    const int gpb = blockDim.x / G;
    const int group = threadIdx.x / G, lig = threadIdx.x - group * G;
    if (group >= gpb) return;
    const int n2 = min(c.num_chunks[1], c.max_chunks2);
    for (int ci = blockIdx.x * gpb + group; ci < n2; ci += gridDim.x * gpb) {
        const int first = c.chunk2_desc[ci * 2 + 0], last = c.chunk2_desc[ci * 2 + 1];
        for (int cv = lig; cv < nvec; cv += G) {
            const int col = cv * V;
            float g[V];
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] = 0.f;
            float q = 0.f;
            sum_partials<V, VEC, 16>(c.partial, c.partial_q, first, last - first, dim, col, g, q);
            if (VEC) stv<V>(c.partial2 + static_cast<size_t>(ci) * dim + col, g);
            if (cv == 0) c.partial2_q[ci] = q;
        }
    }
}

// The optimiser's row-local formula for columns [col, col+V) of one table row, given the row's gradient sum g, its
// scalar q (mean-of-squares sum) and entry count. Shared by the row pass and by the untouched-rows pass below so that
// a row without entries gets bit for bit the same arithmetic (g = 0, q = 0, cnt = 0) from either.
template <int V, int KIND>
struct RowKindTraits {
    static constexpr bool kUsesM = (KIND == ROW_ADAM_MV || KIND == ROW_ADAM_SPARSE_ENT || KIND == ROW_ADAM_DENSE || KIND == ROW_ADAM_FULL);
    static constexpr bool kUsesP = (KIND == ROW_SGD || KIND == ROW_ADAGRAD_ENT || KIND == ROW_ADAM_SPARSE_ENT || KIND == ROW_ADAM_DENSE || KIND == ROW_ADAM_FULL);
};

template <int V, int KIND>
__device__ __forceinline__ void load_row_state(const RowPassArgs& a, size_t off, int cnt, bool p_always,
                                               float (&p)[V], float (&m)[V], float (&v)[V]) {
#pragma unroll
    for (int i = 0; i < V; ++i) { p[i] = 0.f; m[i] = 0.f; v[i] = 0.f; }
    if (RowKindTraits<V, KIND>::kUsesM) { if (a.nt_m) ldv_nt<V>(a.m + off, m); else ldv<V>(a.m + off, m); }
    if (KIND == ROW_ADAM_FULL) ldv<V>(a.v + off, v);
    if (RowKindTraits<V, KIND>::kUsesP) {
        if (p_always || cnt != 0) { if (a.nt_p) ldv_nt<V>(a.P + off, p); else ldv<V>(a.P + off, p); }
    }
}

// lazy decay: the row's P and m first get the factors of the updates the row sat out, one at a time in update order — what
// the dense passes would have done to it (kernels.h). The per-row scalar travels separately (launch_lazy_refresh's
// scalars_only mode leaves an up-to-date snapshot in sc_in).
template <int V, int KIND>
__device__ __forceinline__ void refresh_row_state(const RowPassArgs& a, const float* hist, int64_t row, float (&p)[V], float (&m)[V]) {
    if (!a.pending.stamp) return;
    for (int u = a.pending.stamp[row]; u < a.pending.now; ++u) {
        if (RowKindTraits<V, KIND>::kUsesP) {
            const float d = hist[u % kLazyHistory];
#pragma unroll
            for (int i = 0; i < V; ++i) p[i] *= d;
        }
        if (RowKindTraits<V, KIND>::kUsesM) {
#pragma unroll
            for (int i = 0; i < V; ++i) m[i] *= a.s_m;
        }
    }
}

// sc_old: the row's scalar state going in, for the kinds that have one
template <int V, int KIND>
__device__ __forceinline__ void apply_row_formula_sc(const RowPassArgs& a, int64_t row, bool first_col, size_t off, int cnt,
                                                     bool touch_p, const float (&g)[V], float q, float (&p)[V],
                                                     float (&m)[V], float (&v)[V], float sc_old) {
#pragma clang fp contract(off)      // every product rounded on its own, in whichever kernel this lands (see accumulate_segment)
    if (KIND == ROW_SGD) {
        if (!touch_p) return;
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = p[i] * a.decay + a.lr * g[i];
        if (a.nt_p) stv_nt<V>(a.P + off, p); else stv<V>(a.P + off, p);
    } else if (KIND == ROW_ADAGRAD_ENT) {
        const float acc = sc_old + q;
        if (first_col) a.sc_out[row] = acc;
        if (!touch_p) return;
        const float sc = 1.f / sqrtf(acc + a.eps);
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = p[i] * a.decay + a.lr * (g[i] * sc);
        if (a.nt_p) stv_nt<V>(a.P + off, p); else stv<V>(a.P + off, p);
    } else if (KIND == ROW_SCALAR_ACC) {
        if (first_col) a.sc_out[row] = sc_old + q;
    } else if (KIND == ROW_ADAM_FULL) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float mn = m[i] * a.s_m + a.one_m_b1 * g[i];      // updates_adam.cu:196-200
            mn += (-a.c_reg) * p[i];                         // :203-213
            float ag = g[i] + (-a.lambda) * p[i];            // :264-273
            ag = ag * ag;
            const float vn = v[i] * a.s_v + ag * a.one_m_b2; // :277-281
            m[i] = mn;
            v[i] = vn;
            p[i] = p[i] + ((mn / (sqrtf(vn) + a.eps)) * a.bc) * a.lr;   // :312-328 (λ = 0)
        }
        if (a.nt_m) stv_nt<V>(a.m + off, m); else stv<V>(a.m + off, m);
        stv<V>(a.v + off, v);
        if (a.nt_p) stv_nt<V>(a.P + off, p); else stv<V>(a.P + off, p);
    } else {
        // ROW_ADAM_MV / ROW_ADAM_SPARSE_ENT / ROW_ADAM_DENSE: v is one scalar per row (updates_adam.cu:126).
#pragma unroll
        for (int i = 0; i < V; ++i) m[i] = m[i] * a.s_m + a.one_m_b1 * g[i];
        if (a.nt_m) stv_nt<V>(a.m + off, m); else stv<V>(a.m + off, m);
        const float vn = sc_old * a.s_v + a.one_m_b2 * q;
        if (first_col) a.sc_out[row] = vn;
        if (KIND == ROW_ADAM_SPARSE_ENT) {
            if (!touch_p) return;
            const float denom = sqrtf(vn) + a.eps;
            const float fc = static_cast<float>(cnt);
#pragma unroll
            for (int i = 0; i < V; ++i) p[i] = p[i] * a.decay + (a.lr * fc) * ((a.bc * m[i]) / denom);
            if (a.nt_p) stv_nt<V>(a.P + off, p); else stv<V>(a.P + off, p);
        } else if (KIND == ROW_ADAM_DENSE) {
            const float denom = sqrtf(vn) + a.eps;
#pragma unroll
            for (int i = 0; i < V; ++i) p[i] = p[i] * a.decay + ((m[i] / denom) * a.bc) * a.lr;
            if (a.nt_p) stv_nt<V>(a.P + off, p); else stv<V>(a.P + off, p);
        }
    }
}

template <int KIND>
struct RowKindScalar { static constexpr bool value = (KIND == ROW_ADAGRAD_ENT || KIND == ROW_SCALAR_ACC || KIND == ROW_ADAM_MV ||
                                                      KIND == ROW_ADAM_SPARSE_ENT || KIND == ROW_ADAM_DENSE); };
template <int V, int KIND>
__device__ __forceinline__ void apply_row_formula(const RowPassArgs& a, int64_t row, bool first_col, size_t off, int cnt,
                                                  bool touch_p, const float (&g)[V], float q, float (&p)[V],
                                                  float (&m)[V], float (&v)[V]) {
    float sc_old = 0.f;
    if (RowKindScalar<KIND>::value) sc_old = a.sc_in[row];
    apply_row_formula_sc<V, KIND>(a, row, first_col, off, cnt, touch_p, g, q, p, m, v, sc_old);
}

template <int V, int TABLE, int KIND, int UNROLL>
__global__ __launch_bounds__(256) void row_pass_kernel(Csr c, RowPassArgs a, int G, int nvec) {
    constexpr bool VEC = (KIND != ROW_SCALAR_ACC);
    __shared__ float hist[kLazyHistory];      // lazy decay: the factor history, out of the kernel arguments (per-lane index)
    if (a.pending.stamp) {
        for (int i = threadIdx.x; i < kLazyHistory; i += blockDim.x) hist[i] = a.pending.decay[i];
        __syncthreads();
    }
    const int rpb = blockDim.x / G;
    const int group = threadIdx.x / G, lig = threadIdx.x - group * G;
    if (group >= rpb) return;
    const int dim = a.dim;
    // touched_only: this launch owns the rows of Csr::touched (the rows without entries get their dense decay from
    // untouched_rows_kernel); otherwise every table row
    const int64_t limit = a.touched_only ? static_cast<int64_t>(*c.num_touched) : c.rows;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * rpb + group; r < limit;
         r += static_cast<int64_t>(gridDim.x) * rpb) {
        const int64_t row = a.touched_only ? static_cast<int64_t>(c.touched[r]) : r;
        const int begin = c.row_begin[row], end = c.row_end[row];
        const int cnt = end - begin;
        if (cnt == 0 && !a.dense) continue;
        const bool p_always = (a.decay != 1.f) || KIND == ROW_ADAM_FULL || KIND == ROW_ADAM_DENSE;
        const bool touch_p = p_always || cnt != 0;              // sparse kinds leave untouched rows alone when λ = 0
        for (int cv = lig; cv < nvec; cv += G) {
            const int col = cv * V;
            const size_t off = static_cast<size_t>(row) * dim + col;
            // the row's own state does not depend on the entries: fetch it first so it is in flight during the gather
            float p[V], m[V], v[V];
            load_row_state<V, KIND>(a, off, cnt, p_always, p, m, v);
            refresh_row_state<V, KIND>(a, hist, row, p, m);

            float g[V];
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] = 0.f;
            float q = 0.f;
            if (cnt > c.chunk) {           // long row: ordered sum of its chunk partials
                const int nch = (cnt + c.chunk - 1) / c.chunk;
                if (nch > kFan) sum_partials<V, VEC, 4>(c.partial2, c.partial2_q, c.chunk2_base[row], (nch + kFan - 1) / kFan, dim, col, g, q);
                else sum_partials<V, VEC, 4>(c.partial, c.partial_q, c.chunk_base[row], nch, dim, col, g, q);
            } else if (cnt > 0) {
                accumulate_segment<V, TABLE, VEC, VEC ? UNROLL : kSegUnrollScalar>(a, c.sorted_entry, begin, end, col, g, q);
            }
            apply_row_formula<V, KIND>(a, row, cv == 0, off, cnt, touch_p, g, q, p, m, v);
        }
    }
}

// One launch per pass of a table (chunk_pass_kernel + chunk2_pass_kernel + row_pass_kernel in one grid). The first
// `chunk_blocks` workgroups take the level-1 chunks of the long rows, one per thread group; a group that finishes a
// chunk bumps an arrival counter, and whoever arrives LAST — nobody ever waits — carries on: the last chunk of a level-2
// range sums that range's partials, the last contribution of a row sums the row's partials and applies the optimiser's
// row formula. Which group that is varies from run to run; what it computes does not (the sums run over the partials in
// index order), so results are bit-identical to the three-launch form. The remaining workgroups are the row pass over
// the rows of at most kChunk entries, which depend on nothing: they overlap the chunk work instead of queueing behind
// two kernel boundaries. Counters return to zero (reset by the last arriver) for the next pass.
// Partials in flight in the sums of the chunk tree — the level-2 sum of up to kFan level-1 partials and the final sum of a row
// of up to kFan chunks (the same additions in the same order whatever the depth) — and the kernel's occupancy. Two families,
// measured over every bench shape in separate processes (profiles/r05_exp_table_pass_family.txt):
//   * DEEP (16 / 12 in flight, registers as they come: 110-114, four waves per SIMD) — SGD and Adagrad. Their small-batch steps
//     (LSE recipe, batch 4 096) end in the hottest row's chain of dependent round trips: four in flight in the final sum were
//     sixteen trips behind the row's last chunk (LSE 0.164 -> 0.160 ms with twelve; 0.1591 -> 0.1616 back at eight).
//   * WIDE (8 / 8 in flight, amdgpu_waves_per_eu(5): 85-93 registers, no scratch) — the passes of the Adam modes, whose steps are
//     the rows of at most a chunk's entries, one thread group each, bounded by how many are in flight: batch 6 400 0.2651 ->
//     0.2569 ms, 12 800 0.4015 -> 0.399, 25 600 0.5989 -> 0.5766, 51 200 0.8683 -> 0.8668, full_adam 0.7604 -> 0.7579. (The
//     request on the SGD / Adagrad kinds too: LSE +5 %; on the scalar-only pass: 72 B of scratch, LSE +12 %; eight in flight
//     without the request: full_adam +1.3 %.)
// (Seven or eight source rows in flight in the chunk walk instead of five: 51 200 +1 %, LSE -1 %: not kept.)
constexpr int kL2SumDeep = 16, kRowSumDeep = 12, kL2SumWide = 8, kRowSumWide = 8;
// source rows in flight per lane in the DEEP family's walk of a level-1 chunk (its rows keep SegUnrollDeep): seven — 113-120 registers,
// still four waves per SIMD — against five: Adagrad batch 6 400 0.2388 -> 0.2172 ms, 51 200 0.7635 -> 0.7456, SGD batch 6 400 0.2029
// -> 0.1992, LSE 0.1503 -> 0.1496; six: about the same; eight: 129 registers, three waves, LSE +8 %
template <int TABLE> constexpr int kDeepChunkWalk = 7;
#ifndef NVSM_TABLE_PASS_WAVES
#define NVSM_TABLE_PASS_WAVES 5
#endif
#if NVSM_TABLE_PASS_WAVES > 0
#define NVSM_TABLE_PASS_ATTR __attribute__((amdgpu_waves_per_eu(NVSM_TABLE_PASS_WAVES, NVSM_TABLE_PASS_WAVES)))
#else
#define NVSM_TABLE_PASS_ATTR
#endif
constexpr int kMaxGroupsPerBlock = 256;      // one-column rows: a thread group is a single thread
template <int V, int TABLE, int KIND, int UNROLL, int L2SUM, int ROWSUM, int CHUNKU>
__device__ __forceinline__ void table_pass_body(const Csr& c, const RowPassArgs& a, int G, int nvec, int chunk_blocks) {
    constexpr bool VEC = (KIND != ROW_SCALAR_ACC);
    __shared__ float hist[kLazyHistory];      // lazy decay: the factor history, out of the kernel arguments (per-lane index)
    if (a.pending.stamp) {
        for (int i = threadIdx.x; i < kLazyHistory; i += blockDim.x) hist[i] = a.pending.decay[i];
        __syncthreads();
    }
    const int gpb = blockDim.x / G;
    const int group = threadIdx.x / G, lig = threadIdx.x - group * G;
    const int dim = a.dim;
    if (static_cast<int>(blockIdx.x) < chunk_blocks) {
        __shared__ int todo[kMaxGroupsPerBlock];
        const int nchunks = min(c.num_chunks[0], c.max_chunks);
        // (one turn when the launch has a workgroup per gpb chunks, as the one-launch pass does; the chunk-only launch next to
        //  entry_walk_kernel has a bounded grid and strides)
        // workgroup cb runs on XCD cb % 8 (round-robin dispatch): with a chunk order, XCD x takes the x-th eighth of it
        const int nslots = (nchunks + gpb - 1) / gpb, per_xcd = (nslots + 7) >> 3;
        for (int cb = blockIdx.x; cb < (c.chunk_order ? per_xcd * 8 : nslots); cb += chunk_blocks) {
        const int pos = (c.chunk_order ? (cb & 7) * per_xcd + (cb >> 3) : cb) * gpb + group;
        const bool active = group < gpb && pos < nchunks;
        const int ci = active ? (c.chunk_order ? c.chunk_order[pos] : pos) : 0;
        int row = 0, nch = 0, c_in_row = 0;
        if (active) {
            row = c.chunk_desc[ci * 3 + 0];
            const int begin = c.chunk_desc[ci * 3 + 1], end = c.chunk_desc[ci * 3 + 2];
            nch = (c.row_end[row] - c.row_begin[row] + c.chunk - 1) / c.chunk;
            c_in_row = ci - c.chunk_base[row];
            for (int cv = lig; cv < nvec; cv += G) {
                const int col = cv * V;
                float g[V];
#pragma unroll
                for (int i = 0; i < V; ++i) g[i] = 0.f;
                float q = 0.f;
                accumulate_segment<V, TABLE, VEC, VEC ? CHUNKU : kSegUnrollScalar>(a, c.sorted_entry, begin, end, col, g, q);
                if (VEC) st_agent<V>(c.partial + static_cast<size_t>(ci) * dim + col, g);
                if (cv == 0) st_agent1(c.partial_q + ci, q);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the partial has reached memory before anybody is told
        __syncthreads();
        const bool two_level = nch > kFan;
        if (active && lig == 0) {
            int what = 0;
            if (two_level) {
                const int c2 = c_in_row / kFan;
                const int members = min(kFan, nch - c2 * kFan);
                int* ctr = c.arrive2 + c.chunk2_base[row] + c2;
                if (c.chunk2_base[row] + c2 < c.max_chunks2 &&
                    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                    __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    what = 1;
                }
            } else {
                int* ctr = c.arrive_row + row;
                if (__hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nch - 1) {
                    __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    what = 2;
                }
            }
            todo[group] = what;
        } else if (lig == 0) {
            todo[group] = 0;
        }
        __syncthreads();
        int what = (group < gpb) ? todo[group] : 0;
        __syncthreads();
        // level 2: ordered sum of this range's (up to kFan) level-1 partials
        if (what == 1) {
            const int c2 = c_in_row / kFan;
            const int first = c.chunk_base[row] + c2 * kFan, count = min(kFan, nch - c2 * kFan);
            const int slot = c.chunk2_base[row] + c2;
            for (int cv = lig; cv < nvec; cv += G) {
                const int col = cv * V;
                float g[V];
#pragma unroll
                for (int i = 0; i < V; ++i) g[i] = 0.f;
                float q = 0.f;
                sum_partials_agent<V, VEC, L2SUM>(c.partial, c.partial_q, first, count, dim, col, g, q);
                if (VEC) st_agent<V>(c.partial2 + static_cast<size_t>(slot) * dim + col, g);
                if (cv == 0) st_agent1(c.partial2_q + slot, q);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (what == 1 && lig == 0) {
            const int n2 = (nch + kFan - 1) / kFan;
            int* ctr = c.arrive_row + row;
            int next = 0;
            if (__hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n2 - 1) {
                __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                next = 2;
            }
            todo[group] = next;
        }
        __syncthreads();
        what = (group < gpb) ? todo[group] : 0;
        if (what != 2) continue;
        // the row: ordered sum of its partials, then the row formula (row_pass_kernel's long-row branch)
        const int cnt = c.row_end[row] - c.row_begin[row];
        const bool p_always = (a.decay != 1.f) || KIND == ROW_ADAM_FULL || KIND == ROW_ADAM_DENSE;
        for (int cv = lig; cv < nvec; cv += G) {
            const int col = cv * V;
            const size_t off = static_cast<size_t>(row) * dim + col;
            float p[V], m[V], v[V];
            load_row_state<V, KIND>(a, off, cnt, p_always, p, m, v);
            refresh_row_state<V, KIND>(a, hist, row, p, m);
            float g[V];
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] = 0.f;
            float q = 0.f;
            if (two_level) sum_partials_agent<V, VEC, 4>(c.partial2, c.partial2_q, c.chunk2_base[row], (nch + kFan - 1) / kFan, dim, col, g, q);
            else sum_partials_agent<V, VEC, ROWSUM>(c.partial, c.partial_q, c.chunk_base[row], nch, dim, col, g, q);
            apply_row_formula<V, KIND>(a, row, cv == 0, off, cnt, true, g, q, p, m, v);
        }
        }
        return;
    }
    // ---- rows of at most kChunk entries ----
    if (a.rows_elsewhere || group >= gpb) return;
    const int64_t limit = a.touched_only ? static_cast<int64_t>(*c.num_touched) : c.rows;
    const int64_t nblocks = static_cast<int64_t>(gridDim.x) - chunk_blocks;
    for (int64_t r = (static_cast<int64_t>(blockIdx.x) - chunk_blocks) * gpb + group; r < limit; r += nblocks * gpb) {
        const int64_t row = a.touched_only ? static_cast<int64_t>(c.touched[r]) : r;
        const int begin = c.row_begin[row], end = c.row_end[row];
        const int cnt = end - begin;
        if (cnt > c.chunk || (cnt == 0 && !a.dense)) continue;
        const bool p_always = (a.decay != 1.f) || KIND == ROW_ADAM_FULL || KIND == ROW_ADAM_DENSE;
        const bool touch_p = p_always || cnt != 0;
        for (int cv = lig; cv < nvec; cv += G) {
            const int col = cv * V;
            const size_t off = static_cast<size_t>(row) * dim + col;
            float p[V], m[V], v[V];
            load_row_state<V, KIND>(a, off, cnt, p_always, p, m, v);
            refresh_row_state<V, KIND>(a, hist, row, p, m);
            float g[V];
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] = 0.f;
            float q = 0.f;
            if (cnt > 0) accumulate_segment<V, TABLE, VEC, VEC ? UNROLL : kSegUnrollScalar>(a, c.sorted_entry, begin, end, col, g, q);
            apply_row_formula<V, KIND>(a, row, cv == 0, off, cnt, touch_p, g, q, p, m, v);
        }
    }
}
template <int V, int TABLE, int KIND, int UNROLL>
__global__ __launch_bounds__(256) void table_pass_kernel(Csr c, RowPassArgs a, int G, int nvec, int chunk_blocks) {
    table_pass_body<V, TABLE, KIND, UNROLL, kL2SumDeep, kRowSumDeep, kDeepChunkWalk<TABLE>>(c, a, G, nvec, chunk_blocks);
}
// the passes of the Adam modes (RowPassArgs::wide): see above
template <int V, int TABLE, int KIND, int UNROLL>
__global__ __launch_bounds__(256) NVSM_TABLE_PASS_ATTR void table_pass_wide_kernel(Csr c, RowPassArgs a, int G, int nvec, int chunk_blocks) {
    table_pass_body<V, TABLE, KIND, UNROLL, kL2SumWide, kRowSumWide, SegUnrollWide<TABLE>::value>(c, a, G, nvec, chunk_blocks);
}

// ---- rows of a table much larger than the batch: walk the sorted ENTRIES, not a list of rows --------------------------------
// With about one entry per touched row the list walk above is a chain of four dependent random reads per row (list -> row
// bounds -> entry -> coefficient / source row) with a wave's worth of registers parked on it: 0.8-0.9 ms for the 705 k
// document rows a batch touches at |D| = 2 M, where the bytes moved (3.8 GB) would take 0.6. Here a wave takes 64
// consecutive positions of the sorted (row, entry) arrays — two coalesced loads —, every lane fetches the coefficient / scalar
// terms of its own entry and, if its position opens a row, the row's scalar state; then the wave goes through the rows that
// open in its range one after the other, with everything a row needs (table row, moments, its entries' source rows — ids
// and coefficients come out of the lanes by v_readlane) issued in ONE round of loads. A row whose entries run past the
// wave's 64 positions is finished from the following positions (rare); rows of more than kChunk entries are left to the
// chunk tree (table_pass_kernel finishes those). The sums run over a row's entries in sorted order with
// accumulate_segment's arithmetic: results are bit-identical to the list walk.
template <int TABLE>
__device__ __forceinline__ void entry_terms(const RowPassArgs& a, uint32_t en, bool ok, uint32_t& src, float& cf, float& sq) {
#pragma clang fp contract(off)
    src = static_cast<uint32_t>((static_cast<uint64_t>(en) * a.div_magic) >> 37);
    float c;
    if (TABLE == 0) c = a.wts ? a.wts[en] : 1.f;
    else c = a.coefs ? a.coefs[en] : 1.f;
    const float s = a.sq_src ? a.sq_src[src] : 0.f;
    const float scl = (TABLE == 0 && a.src_scale) ? a.src_scale[src] : 1.f;
    c = ok ? c : 0.f;
    sq = (TABLE == 0) ? c * s : (c * c) * s;
    cf = (TABLE == 0 && a.src_scale) ? c * scl : c;
}

// (source rows of one table row in flight per lane. Four at first; the two entry walks of a step — words and documents — run
//  next to each other for 0.7 ms at |V| = 500 k, |D| = 2 M and, like the row passes of the NVSM shape, do better as a
//  pair when each is less greedy: interleaved A/B 1 / 2 / 3 / 4 / 6 in flight: 1.81 / 1.79 / 1.82 / 1.88 / 1.90 ms per step)
constexpr int kWalkUnroll = 2;
// g, q += the entries held by lanes [first, first + count) of (src, cf, sq), in lane order. NT column turns of 64 lanes each
// are held in registers side by side (a 300-column row is two: 64 + 11 lanes), so that a row still costs ONE round of loads.
template <int V, int NT>
__device__ __forceinline__ void walk_entries(const RowPassArgs& a, uint32_t src, float cf, float sq, int first, int count,
                                             const int (&col)[NT], const bool (&valid)[NT], float (&g)[NT][V], float& q) {
#pragma clang fp contract(off)
    const bool need_q = (a.sq_src != nullptr);
    for (int j = 0; j < count; j += kWalkUnroll) {
        float x[kWalkUnroll][NT][V], cu[kWalkUnroll], su[kWalkUnroll];
#pragma unroll
        for (int u = 0; u < kWalkUnroll; ++u) {
            const bool ok = (j + u) < count;
            const int l = first + (ok ? j + u : count - 1);
            const uint32_t s = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(src), l));
            const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cf), l));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), l));
            cu[u] = ok ? c1 : 0.f;
            su[u] = ok ? s1 : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int i = 0; i < V; ++i) x[u][t][i] = 0.f;
                if (valid[t]) ldv<V>(a.X + static_cast<size_t>(s) * a.dim + col[t], x[u][t]);
            }
        }
#pragma unroll
        for (int u = 0; u < kWalkUnroll; ++u) {
            if (need_q) q += su[u];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int i = 0; i < V; ++i) g[t][i] = __builtin_fmaf(cu[u], x[u][t][i], g[t][i]);
            }
        }
    }
}

template <int V, int TABLE, int KIND, int NT>
__global__ __launch_bounds__(256) void entry_walk_kernel(Csr c, RowPassArgs a, int nvec, int ppw) {
    __shared__ float hist[kLazyHistory];
    if (a.pending.stamp) {
        for (int i = threadIdx.x; i < kLazyHistory; i += blockDim.x) hist[i] = a.pending.decay[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int dim = a.dim;
    const int64_t n = c.n;
    int col[NT]; bool valid[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { valid[t] = (t * 64 + lane) < nvec; col[t] = (valid[t] ? t * 64 + lane : 0) * V; }
    // ppw (a power of two <= 64): sorted positions per wave. 64 when the batch has waves to spare; a small batch gets short
    // ranges so that its rows spread over the machine instead of queueing fifty to a wave (lanes >= ppw then only work on columns)
    const int64_t nblocks = (n + ppw - 1) / ppw;
    const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    for (int64_t blk = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; blk < nblocks; blk += nwaves) {
        const int64_t base = blk * ppw;
        const int64_t i = base + lane;
        const bool in = lane < ppw && i < n;
        const int key = in ? c.sorted_key[i] : -1;
        const int prev = (in && i > 0) ? c.sorted_key[i - 1] : -1;
        const uint32_t en = in ? static_cast<uint32_t>(c.sorted_entry[i]) : 0u;
        const int next_key = (base + ppw < n) ? c.sorted_key[base + ppw] : -1;          // wave-uniform
        uint32_t src; float cf, sq;
        entry_terms<TABLE>(a, en, in, src, cf, sq);
        const bool head = in && key != prev;
        float sc_mine = 0.f;
        if (RowKindScalar<KIND>::value && head) sc_mine = a.sc_in[key];
        uint64_t heads = __ballot(head);
        while (heads) {
            const int h = __ffsll(static_cast<long long>(heads)) - 1;
            heads &= heads - 1;
            const int row = __builtin_amdgcn_readlane(key, h);
            const int cnt1 = __popcll(__ballot(key == row));                           // lanes h .. h + cnt1 - 1 (sorted)
            int cnt2 = 0;
            uint32_t src2 = 0; float cf2 = 0.f, sq2 = 0.f;
            if (h + cnt1 == ppw && next_key == row) {                                   // the row runs on past this wave's range
                const int64_t i2 = base + ppw + lane;
                const bool in2 = i2 < n;
                const int key2 = in2 ? c.sorted_key[i2] : -1;
                const uint32_t en2 = in2 ? static_cast<uint32_t>(c.sorted_entry[i2]) : 0u;
                entry_terms<TABLE>(a, en2, in2, src2, cf2, sq2);
                cnt2 = __popcll(__ballot(key2 == row));
            }
            const int cnt = cnt1 + cnt2;
            if (cnt > c.chunk) continue;                                                 // chunk tree (table_pass_kernel)
            const float sc_old = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc_mine), h));
            float p[NT][V], m[NT][V], v[NT][V], g[NT][V];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int k = 0; k < V; ++k) { p[t][k] = 0.f; m[t][k] = 0.f; v[t][k] = 0.f; g[t][k] = 0.f; }
                if (valid[t]) load_row_state<V, KIND>(a, static_cast<size_t>(row) * dim + col[t], cnt, true, p[t], m[t], v[t]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) { if (valid[t]) refresh_row_state<V, KIND>(a, hist, row, p[t], m[t]); }
            float q = 0.f;
            walk_entries<V, NT>(a, src, cf, sq, h, cnt1, col, valid, g, q);
            if (cnt2) walk_entries<V, NT>(a, src2, cf2, sq2, 0, cnt2, col, valid, g, q);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (valid[t]) apply_row_formula_sc<V, KIND>(a, row, t == 0 && lane == 0, static_cast<size_t>(row) * dim + col[t], cnt, true,
                                                            g[t], q, p[t], m[t], v[t], sc_old);
            }
        }
    }
}

// Rows without entries, when the pass is dense (λ > 0 and / or Adam's m·β₁, v·β₂ decay): most rows of a table much
// larger than the batch. The row pass spends one wave and two dependent round trips (bounds, then state) on such a
// row and its registers are sized for the 8-deep gather (3 waves per SIMD): 3 TB/s at |D| = 2M. This pass streams
// them instead — one (row, 16 B column group) item per thread, four independent items in flight per thread, few
// registers — and leaves the rows that do have entries to the row pass (launched on Csr::touched only).
constexpr int kUntouchedUnroll = 4;      // (2: the same; 8: LSE +4 %)
template <int V, int KIND>
__global__ __launch_bounds__(256) void untouched_rows_kernel(Csr c, RowPassArgs a, uint32_t nvec, uint64_t total) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const bool p_always = (a.decay != 1.f);
    for (uint64_t base = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; base < total;
         base += stride * kUntouchedUnroll) {
        int64_t row[kUntouchedUnroll]; uint32_t cv[kUntouchedUnroll]; bool mine[kUntouchedUnroll];
#pragma unroll
        for (int u = 0; u < kUntouchedUnroll; ++u) {
            const uint64_t q = base + u * stride;
            const uint64_t qq = q < total ? q : total - 1;
            row[u] = static_cast<int64_t>(qq / nvec);
            cv[u] = static_cast<uint32_t>(qq - static_cast<uint64_t>(row[u]) * nvec);
            mine[u] = (q < total) && (c.row_end[row[u]] == c.row_begin[row[u]]);
        }
        float p[kUntouchedUnroll][V], m[kUntouchedUnroll][V], v[kUntouchedUnroll][V];
#pragma unroll
        for (int u = 0; u < kUntouchedUnroll; ++u)
            if (mine[u]) load_row_state<V, KIND>(a, static_cast<size_t>(row[u]) * a.dim + cv[u] * V, 0, p_always, p[u], m[u], v[u]);
#pragma unroll
        for (int u = 0; u < kUntouchedUnroll; ++u) {
            if (!mine[u]) continue;
            float g[V];
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] = 0.f;
            apply_row_formula<V, KIND>(a, row[u], cv[u] == 0, static_cast<size_t>(row[u]) * a.dim + cv[u] * V, 0, p_always,
                                       g, 0.f, p[u], m[u], v[u]);
        }
    }
}

// One wave per row (every lane reads the row's stamp before lane 0 rewrites it): P, m of the row and its scalar state get
// the factors of the updates (stamp, now] applied one at a time — the roundings of the dense pass, see kernels.h.
template <int V>
__global__ __launch_bounds__(256) void lazy_refresh_kernel(LazyRefreshArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    const int64_t limit = a.list ? static_cast<int64_t>(*a.list_count) : a.rows;
    for (int64_t it = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; it < limit; it += waves) {
        const int64_t row = a.list ? static_cast<int64_t>(a.list[it]) : it;
        // (wave-uniform, in a scalar register: the factor below is then a scalar load from the kernel arguments instead of a
        //  per-lane index into a private copy of the array; never further back than the history reaches)
        const int from = max(__builtin_amdgcn_readfirstlane(a.stamp[row]), a.now - kLazyHistory);
        if (from != a.now) {
            for (int c = lane * V; c < a.dim; c += 64 * V) {
                const size_t off = static_cast<size_t>(row) * a.dim + c;
                float p[V], m[V];
                ldv<V>(a.P + off, p);
                if (a.m) ldv<V>(a.m + off, m);
                for (int u = from; u < a.now; ++u) {
                    const float d = a.decay[u % kLazyHistory];
#pragma unroll
                    for (int i = 0; i < V; ++i) p[i] *= d;
                    if (a.m) {
#pragma unroll
                        for (int i = 0; i < V; ++i) m[i] *= a.s_m;
                    }
                }
                stv<V>(a.P + off, p);
                if (a.m) stv<V>(a.m + off, m);
            }
        }
        if (lane == 0) {
            if (a.sc) {
                float v = a.sc[row];
                if (a.s_v != 1.f) for (int u = from; u < a.now; ++u) v *= a.s_v;
                if (from != a.now) a.sc[row] = v;
                if (a.sc_snapshot) a.sc_snapshot[row] = v;
            }
            a.stamp[row] = a.now;
        }
    }
}

// scalars_only: one thread per listed row; the scalar is brought up to date in place and snapshotted, nothing else moves
__global__ void lazy_scalar_snapshot_kernel(LazyRefreshArgs a) {
    const int64_t limit = a.list ? static_cast<int64_t>(*a.list_count) : a.rows;
    for (int64_t it = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; it < limit;
         it += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t row = a.list ? static_cast<int64_t>(a.list[it]) : it;
        const int from = a.stamp[row];
        float v = a.sc[row];
        if (a.s_v != 1.f) for (int u = from; u < a.now; ++u) v *= a.s_v;
        // (the in-place value stays as it was: the row pass, which reads the snapshot, writes the new one over it and then
        //  stamps the row; a row's pending count therefore still describes what is stored)
        a.sc_snapshot[row] = v;
    }
}

// After the last pass of a lazy table's update: the rows the batch touched now carry this update. A launch of its own,
// not a store at the end of the row pass: a row's thread group may straddle two waves (G = 75 lanes for 300 columns, 3
// for 12), and the second wave must still find the old stamp when it reads what the row sat out.
__global__ void stamp_rows_kernel(const int* __restrict__ list, const int* __restrict__ count, int* __restrict__ stamp, int value) {
    const int64_t limit = *count;
    for (int64_t it = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; it < limit;
         it += static_cast<int64_t>(gridDim.x) * blockDim.x) stamp[list[it]] = value;
}
void launch_stamp_rows(const Csr& c, int* stamp, int value, int64_t max_rows, hipStream_t s) {
    if (max_rows <= 0) return;
    hipLaunchKernelGGL(stamp_rows_kernel, dim3(stream_grid(max_rows, 256)), dim3(256), 0, s, c.touched, c.num_touched, stamp, value);
}

void launch_lazy_refresh(const LazyRefreshArgs& a, int64_t max_rows, hipStream_t s) {
    if (max_rows <= 0) return;
    if (a.scalars_only) {
        if (a.sc && a.sc_snapshot) hipLaunchKernelGGL(lazy_scalar_snapshot_kernel, dim3(stream_grid(max_rows, 256)), dim3(256), 0, s, a);
        return;
    }
    int64_t blocks = (max_rows + 3) / 4;                       // 4 waves per workgroup, one row per wave
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (a.dim % 4 == 0) hipLaunchKernelGGL(lazy_refresh_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lazy_refresh_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
}

// Touched-row list + streaming pass over the rest when the table has at least half as many rows as the batch has entries
// (then at least 14 % of the rows are without entries for uniform ids, most of them for Zipf ids).
// NVSM_SPLIT_RATIO overrides the factor (experiments); the lazy decay of model.cpp uses the same rule.
double table_split_ratio() {
    return tuning().split_ratio;
}
bool row_pass_split(const Csr& c) { return c.n > 0 && static_cast<double>(c.rows) * table_split_ratio() >= static_cast<double>(c.n); }
static bool kind_is_row_local_when_untouched(int kind) { return kind != ROW_ADAM_DENSE && kind != ROW_ADAM_FULL; }

static void group_geometry(const RowPassArgs& a, int& V, int& nvec, int& G) {
    // (ROW_SCALAR_ACC, where only the row's scalar moves, was tried with one THREAD per row instead of a thread group as wide
    //  as the row: 64 rows of different lengths per wave run as long as the longest — 36 instead of 27 us for the Adagrad
    //  accumulator pass of the LSE recipe)
    V = (a.dim % 4 == 0) ? 4 : 1;
    nvec = a.dim / V;
    G = nvec < 256 ? nvec : 256;
}

template <int V, int TABLE>
static void chunk_pass_dispatch(const Csr& c, const RowPassArgs& a, int G, int nvec, hipStream_t s) {
    const int gpb = 256 / G;
    int grid = (c.max_chunks + gpb - 1) / gpb;
    int grid2 = (c.max_chunks2 + gpb - 1) / gpb;
    // (the kernels grid-stride over the chunks actually in use) NVSM_CHUNK_GRID_CAP: experiments
    const int cap = tuning().chunk_grid_cap;
    if (cap > 0 && TABLE == 1) { grid = grid < cap ? grid : cap; grid2 = grid2 < cap ? grid2 : cap; }
    if (a.kind == ROW_SCALAR_ACC) {
        hipLaunchKernelGGL((chunk_pass_kernel<V, TABLE, false>), dim3(grid), dim3(256), 0, s, c, a, G, nvec);
        hipLaunchKernelGGL((chunk2_pass_kernel<V, false>), dim3(grid2), dim3(256), 0, s, c, a.dim, G, nvec);
    } else {
        hipLaunchKernelGGL((chunk_pass_kernel<V, TABLE, true>), dim3(grid), dim3(256), 0, s, c, a, G, nvec);
        hipLaunchKernelGGL((chunk2_pass_kernel<V, true>), dim3(grid2), dim3(256), 0, s, c, a.dim, G, nvec);
    }
}

void launch_chunk_pass(const Csr& c, const RowPassArgs& a, hipStream_t s) {
    if (c.n <= 0 || c.max_chunks <= 0) return;
    int V, nvec, G;
    group_geometry(a, V, nvec, G);
    if (V == 4) { if (a.table == 0) chunk_pass_dispatch<4, 0>(c, a, G, nvec, s); else chunk_pass_dispatch<4, 1>(c, a, G, nvec, s); }
    else        { if (a.table == 0) chunk_pass_dispatch<1, 0>(c, a, G, nvec, s); else chunk_pass_dispatch<1, 1>(c, a, G, nvec, s); }
}

template <int V, int TABLE>
static void row_pass_dispatch(const Csr& c, const RowPassArgs& a, int G, int nvec, hipStream_t s) {
    const int rpb = 256 / G;
    int64_t blocks = (c.rows + rpb - 1) / rpb;
    if (blocks > 256 * 64) blocks = 256 * 64;
    // a.max_blocks > 0: persistent grid-stride launch that leaves room on every CU for a concurrent kernel
    if (a.max_blocks > 0 && blocks > a.max_blocks) blocks = a.max_blocks;
    const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define NVSM_ROW_CASE(K) case K: \
        if (a.shallow) hipLaunchKernelGGL((row_pass_kernel<V, TABLE, K, kSegUnrollShallow>), grid, block, 0, s, c, a, G, nvec); \
        else hipLaunchKernelGGL((row_pass_kernel<V, TABLE, K, SegUnrollDeep<TABLE>::value>), grid, block, 0, s, c, a, G, nvec); \
        break;
    switch (a.kind) {
        NVSM_ROW_CASE(ROW_SGD)
        NVSM_ROW_CASE(ROW_ADAGRAD_ENT)
        NVSM_ROW_CASE(ROW_ADAM_MV)
        NVSM_ROW_CASE(ROW_ADAM_SPARSE_ENT)
        NVSM_ROW_CASE(ROW_ADAM_DENSE)
        NVSM_ROW_CASE(ROW_ADAM_FULL)
        NVSM_ROW_CASE(ROW_SCALAR_ACC)
        default: break;
    }
#undef NVSM_ROW_CASE
}

template <int V>
static void untouched_dispatch(const Csr& c, const RowPassArgs& a, int nvec, hipStream_t s) {
    const uint32_t nv = (a.kind == ROW_SCALAR_ACC) ? 1u : static_cast<uint32_t>(nvec);      // only the row scalar moves
    const uint64_t total = static_cast<uint64_t>(c.rows) * nv;
    const uint64_t per_block = 256ull * kUntouchedUnroll;
    uint64_t blocks = (total + per_block - 1) / per_block;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define NVSM_UNT_CASE(K) case K: hipLaunchKernelGGL((untouched_rows_kernel<V, K>), grid, block, 0, s, c, a, nv, total); break;
    switch (a.kind) {
        NVSM_UNT_CASE(ROW_SGD)
        NVSM_UNT_CASE(ROW_ADAGRAD_ENT)
        NVSM_UNT_CASE(ROW_ADAM_MV)
        NVSM_UNT_CASE(ROW_ADAM_SPARSE_ENT)
        NVSM_UNT_CASE(ROW_SCALAR_ACC)
        default: break;
    }
#undef NVSM_UNT_CASE
}

void launch_row_pass(const Csr& c, const RowPassArgs& a_in, hipStream_t s, hipStream_t untouched_s) {
    if (!untouched_s) untouched_s = s;
    if (c.rows <= 0) return;
    int V, nvec, G;
    group_geometry(a_in, V, nvec, G);
    RowPassArgs a = a_in;
    a.touched_only = 0; a.shallow = 0;
    Csr cc = c;
    // Table much larger than the batch (at most one entry per row on average): the rows with entries go through the
    // row pass by list, all the others — if the pass is dense — through the streaming pass.
    if (row_pass_split(c) && kind_is_row_local_when_untouched(a.kind)) {
        if (a.dense && !a.lazy && !a.untouched_done) { if (V == 4) untouched_dispatch<4>(c, a, nvec, untouched_s); else untouched_dispatch<1>(c, a, nvec, untouched_s); }
        a.touched_only = 1;
        a.shallow = c.rows >= c.n;      // at most one entry per row on average
        cc.rows = c.n < c.rows ? c.n : c.rows;      // upper bound of the list length: sizes the grid
    }
    if (V == 4) { if (a.table == 0) row_pass_dispatch<4, 0>(cc, a, G, nvec, s); else row_pass_dispatch<4, 1>(cc, a, G, nvec, s); }
    else        { if (a.table == 0) row_pass_dispatch<1, 0>(cc, a, G, nvec, s); else row_pass_dispatch<1, 1>(cc, a, G, nvec, s); }
}

bool launch_untouched_rows(const Csr& c, const RowPassArgs& a, hipStream_t s) {
    if (c.rows <= 0 || !(row_pass_split(c) && kind_is_row_local_when_untouched(a.kind)) || !a.dense || a.lazy) return false;
    int V, nvec, G;
    group_geometry(a, V, nvec, G);
    if (V == 4) untouched_dispatch<4>(c, a, nvec, s); else untouched_dispatch<1>(c, a, nvec, s);
    return true;
}

// NVSM_MERGED_PASS=0 (A/B runs, tests): the three-launch form
static bool& merged_pass_flag() {      // (process-wide test hook: nvsm_debug_set_table_pass_form)
    static bool on = true;
    return on;
}
static bool merged_pass_enabled() { return merged_pass_flag() && tuning().merged_pass; }
void set_table_pass_one_launch(bool on) { merged_pass_flag() = on; }

template <int V, int TABLE>
static void table_pass_dispatch(const Csr& c, const RowPassArgs& a, int G, int nvec, int64_t row_items, hipStream_t s) {
    const int gpb = 256 / G;
    int chunk_blocks = (c.n > 0 && c.max_chunks > 0) ? (c.max_chunks + gpb - 1) / gpb + 8 : 0;      // (+ 8: an eighth per XCD, rounded up)
    // (at most 2048 chunk workgroups — a multiple of 8: workgroup cb runs on XCD cb % 8 —, which stride over the chunks in
    //  use; one per gpb chunk SLOTS, most of them empty, was 5300: 0.994 -> 0.989 ms per step; 256: 1.06. NVSM_CHUNK_BLOCKS overrides)
    const int cb_cap = tuning().chunk_blocks;
    if (cb_cap > 0 && chunk_blocks > cb_cap) chunk_blocks = cb_cap;
    if (a.rows_elsewhere && chunk_blocks > 2048) chunk_blocks = 2048;      // normally there is no chunk at all: a launch that costs nothing
    int64_t row_blocks = (row_items + gpb - 1) / gpb;
    // (32 workgroups per CU, the rest by striding: 16 384 -> 8 192: NVSM shape 1.001 -> 0.991 ms per step, adagrad 0.872 ->
    //  0.851; 4 096: 1.02; 65 536: 1.018 — NVSM_ROW_BLOCKS_CAP for experiments)
    { const int cap = tuning().row_blocks_cap; if (row_blocks > cap) row_blocks = cap; }
    if (a.max_blocks > 0 && row_blocks > a.max_blocks) row_blocks = a.max_blocks;
    if (row_blocks < 1) row_blocks = 1;
    if (a.rows_elsewhere) row_blocks = 0;
    if (chunk_blocks + row_blocks < 1) return;
    const dim3 grid(static_cast<unsigned>(chunk_blocks + row_blocks)), block(256);
    // (NVSM_LAUNCH: a pass the caller times carries the event pair as its own start / stop — not the chunk-only launch in front of
    //  an entry walk, which leaves them to the walk)
#define NVSM_TABLE_LAUNCH(KERNEL, K, DEPTH) \
        if (a.rows_elsewhere) { \
            if (a.shallow) hipLaunchKernelGGL((KERNEL<V, TABLE, K, kSegUnrollShallow>), grid, block, 0, s, c, a, G, nvec, chunk_blocks); \
            else hipLaunchKernelGGL((KERNEL<V, TABLE, K, DEPTH<TABLE>::value>), grid, block, 0, s, c, a, G, nvec, chunk_blocks); \
        } else if (a.shallow) NVSM_LAUNCH((KERNEL<V, TABLE, K, kSegUnrollShallow>), grid, block, 0, s, c, a, G, nvec, chunk_blocks); \
        else NVSM_LAUNCH((KERNEL<V, TABLE, K, DEPTH<TABLE>::value>), grid, block, 0, s, c, a, G, nvec, chunk_blocks);
#define NVSM_TABLE_CASE(K) case K: NVSM_TABLE_LAUNCH(table_pass_kernel, K, SegUnrollDeep) break;
#define NVSM_TABLE_CASE_WIDE(K) case K: NVSM_TABLE_LAUNCH(table_pass_wide_kernel, K, SegUnrollWide) break;
    // (ROW_SGD is both: the last pass of sparse Adam's words update and the whole of SGD's)
    if (a.wide && a.kind == ROW_SGD) { NVSM_TABLE_LAUNCH(table_pass_wide_kernel, ROW_SGD, SegUnrollWide) return; }
    switch (a.kind) {
        NVSM_TABLE_CASE(ROW_SGD)
        NVSM_TABLE_CASE(ROW_ADAGRAD_ENT)
        NVSM_TABLE_CASE_WIDE(ROW_ADAM_MV)
        NVSM_TABLE_CASE_WIDE(ROW_ADAM_SPARSE_ENT)
        NVSM_TABLE_CASE_WIDE(ROW_ADAM_DENSE)
        NVSM_TABLE_CASE_WIDE(ROW_ADAM_FULL)
        NVSM_TABLE_CASE(ROW_SCALAR_ACC)
        default: break;
    }
#undef NVSM_TABLE_CASE_WIDE
#undef NVSM_TABLE_LAUNCH
#undef NVSM_TABLE_CASE
}

// ---- level-1 chunks in batch order --------------------------------------------------------------------------------------
// A chunk is 64 consecutive entries of one hot row, and a row's entries are in batch order (stable sort): chunk c of a row
// that sits in a fraction p of the windows covers about 64 / p consecutive windows. The chunks of DIFFERENT hot rows over
// the same windows gather the same source rows (a window's gradient row goes to every word of the window: 4.4 of a
// window's 10 words are among the 88 hottest at Zipf(1)), but numbered row by row they run at different times on
// different XCDs, whose L2s do not talk to each other: every one of those reads misses. Ordered by the batch position of
// their first entry, with each XCD given one eighth of the batch (table_pass_kernel), they meet in one L2 within
// microseconds of each other. Same chunks, same sums: results do not change.
__global__ void chunk_order_key_kernel(Csr c, int* __restrict__ key) {
    const int nchunks = min(c.num_chunks[0], c.max_chunks);
    if (nchunks == 0) return;      // (no chunks: their order is not looked at, and the sort behind this kernel returns at once too)
    for (int ci = blockIdx.x * blockDim.x + threadIdx.x; ci < c.max_chunks; ci += gridDim.x * blockDim.x) {
        int k = 511;                                                           // unused slots sort behind the chunks (stable)
        if (ci < nchunks) {
            const uint64_t e = static_cast<uint32_t>(c.sorted_entry[c.chunk_desc[ci * 3 + 1]]);
            k = static_cast<int>(min<uint64_t>(511, e * 512 / static_cast<uint64_t>(c.n)));
        }
        key[ci] = k;
    }
}
void launch_chunk_order(const Csr& c, int* key_in, int* key_out, void* sort_temp, size_t sort_temp_bytes, hipStream_t s, bool keys_written) {
    if (!c.chunk_order || c.n <= 0 || c.max_chunks <= 0) return;
    if (!keys_written) hipLaunchKernelGGL(chunk_order_key_kernel, dim3(stream_grid(c.max_chunks, 256)), dim3(256), 0, s, c, key_in);
    sort_pairs(sort_temp, sort_temp_bytes, nullptr, key_in, key_out, nullptr, c.chunk_order, c.max_chunks, 9, nullptr, s, nullptr, 0, c.num_chunks);
}

// NVSM_ENTRY_WALK=0 (A/B runs, tests): the list walk for the rows of tables much larger than the batch
static bool entry_walk_enabled() {
    return tuning().entry_walk;
}
// NVSM_ENTRY_WALK_MIN (tests: 0 sends small batches through the entry walk too)
static int64_t entry_walk_min_entries(int table) {
    const Tuning& t = tuning();
    const long long per_table = table == 0 ? t.entry_walk_min_words : t.entry_walk_min_docs;      // experiments: per table
    return per_table >= 0 ? per_table : t.entry_walk_min;
}
static bool entry_walk_kind(int kind) { return kind == ROW_SGD || kind == ROW_ADAGRAD_ENT || kind == ROW_ADAM_MV || kind == ROW_ADAM_SPARSE_ENT; }

template <int V, int TABLE>
static void entry_walk_dispatch(const Csr& c, const RowPassArgs& a, int nvec, hipStream_t s) {
    if (c.n <= 0) return;
    // sorted positions per wave: enough waves to fill the machine, and at most sixteen — a wave goes through the rows that
    // open in its range one after the other, and with 64 positions (≈50 rows at one entry per row) it was a chain of fifty
    // dependent rounds of loads: 64 / 32 / 16 / 8 / 4 positions: 1.834 / 1.80 / 1.75 / 1.755 / 1.79 ms per step at |D| = 2 M
    int ppw = 16;
    while (ppw > 4 && c.n / ppw < 8192) ppw >>= 1;
    int64_t blocks = ((c.n + ppw - 1) / ppw + 3) / 4;
    if (blocks > 256 * 64) blocks = 256 * 64;
    const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define NVSM_WALK_CASE(K) case K: \
        if (nvec <= 64) NVSM_LAUNCH((entry_walk_kernel<V, TABLE, K, 1>), grid, block, 0, s, c, a, nvec, ppw); \
        else NVSM_LAUNCH((entry_walk_kernel<V, TABLE, K, 2>), grid, block, 0, s, c, a, nvec, ppw); \
        break;
    switch (a.kind) {
        NVSM_WALK_CASE(ROW_SGD)
        NVSM_WALK_CASE(ROW_ADAGRAD_ENT)
        NVSM_WALK_CASE(ROW_ADAM_MV)
        NVSM_WALK_CASE(ROW_ADAM_SPARSE_ENT)
        default: break;
    }
#undef NVSM_WALK_CASE
}

// one pass over a table: chunk tree of the long rows + row formula, in one launch (table_pass_kernel)
int launch_table_pass(const Csr& c, const RowPassArgs& a_in, hipStream_t s, hipStream_t untouched_s) {
    if (!untouched_s) untouched_s = s;
    const bool split = row_pass_split(c) && kind_is_row_local_when_untouched(a_in.kind);
    if (!merged_pass_enabled()) {
        launch_chunk_pass(c, a_in, s); launch_row_pass(c, a_in, s, untouched_s);
        return split ? TABLE_PASS_LIST_WALK : TABLE_PASS_DENSE;
    }
    if (c.rows <= 0) return TABLE_PASS_DENSE;
    int V, nvec, G;
    group_geometry(a_in, V, nvec, G);
    RowPassArgs a = a_in;
    a.touched_only = 0; a.shallow = 0;
    int64_t row_items = c.rows;
    if (row_pass_split(c) && kind_is_row_local_when_untouched(a.kind)) {      // as launch_row_pass
        if (a.dense && !a.lazy && !a.untouched_done) { if (V == 4) untouched_dispatch<4>(c, a, nvec, untouched_s); else untouched_dispatch<1>(c, a, nvec, untouched_s); }
        a.touched_only = 1;
        a.shallow = c.rows >= c.n;
        row_items = c.n < c.rows ? c.n : c.rows;
    }
    // (rows of up to two waves' width, held side by side in registers; batches of a few thousand windows are a chain of launch latencies, not of round trips per
    //  row, and keep the one-launch list walk: LSE batch 4096 0.234 vs 0.254 ms per step)
    if (a.touched_only && nvec <= 128 && c.n >= entry_walk_min_entries(a.table) && entry_walk_enabled() && entry_walk_kind(a.kind)) {
        // the rows with entries, by walking the sorted entries; the chunk tree (if the batch can have rows that long) in a
        // launch of its own, which also finishes those rows
        a.rows_elsewhere = 1;
        if (c.max_chunks > 0) {
            if (V == 4) { if (a.table == 0) table_pass_dispatch<4, 0>(c, a, G, nvec, 0, s); else table_pass_dispatch<4, 1>(c, a, G, nvec, 0, s); }
            else        { if (a.table == 0) table_pass_dispatch<1, 0>(c, a, G, nvec, 0, s); else table_pass_dispatch<1, 1>(c, a, G, nvec, 0, s); }
        }
        if (V == 4) { if (a.table == 0) entry_walk_dispatch<4, 0>(c, a, nvec, s); else entry_walk_dispatch<4, 1>(c, a, nvec, s); }
        else        { if (a.table == 0) entry_walk_dispatch<1, 0>(c, a, nvec, s); else entry_walk_dispatch<1, 1>(c, a, nvec, s); }
        return TABLE_PASS_ENTRY_WALK;
    }
    if (V == 4) { if (a.table == 0) table_pass_dispatch<4, 0>(c, a, G, nvec, row_items, s); else table_pass_dispatch<4, 1>(c, a, G, nvec, row_items, s); }
    else        { if (a.table == 0) table_pass_dispatch<1, 0>(c, a, G, nvec, row_items, s); else table_pass_dispatch<1, 1>(c, a, G, nvec, row_items, s); }
    return a.touched_only ? TABLE_PASS_LIST_WALK : TABLE_PASS_DENSE;
}

// =============================================================================================
// window > 1 per-example passes
// =============================================================================================
// scale[b] = 1 / sqrt(mean_j acc[idx[b,j]] + ε)      (adagrad_update_kernel, cpp/updates_adagrad.cu:83-97)
__global__ void adagrad_scale_kernel(const float* __restrict__ acc, const int* __restrict__ idx, int window, int64_t B,
                                     float eps, float* __restrict__ scale) {
    constexpr int U = 4;      // ids, then accumulators, four in flight (summed in window order)
    for (int64_t b = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; b < B;
         b += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        float s = 0.f;
        for (int j0 = 0; j0 < window; j0 += U) {
            int r[U]; float a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = idx[b * window + min(j0 + u, window - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = acc[r[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) if (j0 + u < window) s += a[u];
        }
        s /= static_cast<float>(window);
        scale[b] = 1.f / sqrtf(s + eps);
    }
}
void launch_adagrad_scale(const float* acc, const int* idx, int window, int64_t B, float eps, float* scale, hipStream_t s) {
    if (B > 0) hipLaunchKernelGGL(adagrad_scale_kernel, dim3(stream_grid(B, 256)), dim3(256), 0, s, acc, idx, window, B, eps, scale);
}

// U[b] = bc · mean_j m[idx[b,j]] / (sqrt(mean_j v[idx[b,j]]) + ε)      (adam_sparse_update_kernel, cpp/updates_adam.cu:132-151)
// (U rows of the window in flight per lane, ids first — see gather_mean_kernel; sums in window order)
template <int V, int U>
__global__ __launch_bounds__(256) void adam_u_kernel(const float* __restrict__ m, const float* __restrict__ v, int dim,
                                                     const int* __restrict__ idx, int window, uint32_t total, uint32_t nvec,
                                                     float bc, float eps, float* __restrict__ Uout) {
    const float fw = static_cast<float>(window);
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const uint32_t b = q / nvec;
        const uint32_t c = (q - b * nvec) * V;
        const int* ip = idx + static_cast<size_t>(b) * window;
        float am[V];
#pragma unroll
        for (int i = 0; i < V; ++i) am[i] = 0.f;
        float av = 0.f;
        for (int j0 = 0; j0 < window; j0 += U) {
            size_t row[U]; float x[U][V], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) row[u] = static_cast<size_t>(ip[min(j0 + u, window - 1)]);
#pragma unroll
            for (int u = 0; u < U; ++u) { ldv<V>(m + row[u] * dim + c, x[u]); vv[u] = v[row[u]]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j0 + u < window) {
#pragma unroll
                    for (int i = 0; i < V; ++i) am[i] += x[u][i];
                    av += vv[u];
                }
            }
        }
        av /= fw;
        const float denom = sqrtf(av) + eps;
#pragma unroll
        for (int i = 0; i < V; ++i) am[i] = bc * (am[i] / fw) / denom;
        stv<V>(Uout + static_cast<size_t>(b) * dim + c, am);
    }
}
template <int V>
static void adam_u_dispatch(int U, dim3 grid, hipStream_t s, const float* m, const float* v, int dim, const int* idx, int window,
                            uint32_t total, uint32_t nvec, float bc, float eps, float* Uout) {
#define NVSM_ADAMU_CASE(N) case N: hipLaunchKernelGGL((adam_u_kernel<V, N>), grid, dim3(256), 0, s, m, v, dim, idx, window, total, nvec, bc, eps, Uout); break;
    switch (U) {
        NVSM_ADAMU_CASE(1) NVSM_ADAMU_CASE(2) NVSM_ADAMU_CASE(3) NVSM_ADAMU_CASE(4) NVSM_ADAMU_CASE(5)
        NVSM_ADAMU_CASE(6) NVSM_ADAMU_CASE(7) NVSM_ADAMU_CASE(8) NVSM_ADAMU_CASE(9) NVSM_ADAMU_CASE(10)
        default: break;
    }
#undef NVSM_ADAMU_CASE
}
void launch_adam_u(const float* m, const float* v, int dim, const int* idx, int window, int64_t B, float bc, float eps,
                   float* U, hipStream_t s) {
    if (B <= 0) return;
    const bool vec = dim % 4 == 0;
    const uint32_t nvec = vec ? dim / 4 : dim, total = static_cast<uint32_t>(B * nvec);
    const dim3 grid(stream_grid(total, 256));
    const int wu = window_unroll(window);      // (five / seven in flight instead of the window's ten, as the word gather now does: nothing / +1 %)
    if (vec) adam_u_dispatch<4>(wu, grid, s, m, v, dim, idx, window, total, nvec, bc, eps, U);
    else adam_u_dispatch<1>(wu, grid, s, m, v, dim, idx, window, total, nvec, bc, eps, U);
}

// =============================================================================================
// Dense projection optimiser: one launch over the nT + nb scalars of (T, b).
// Quirks reproduced: Adam folds L2 into the gradient for T only
// (include/cuNVSM/updates.h:39-62); the bias slot of every TransformStorage::update hard-codes λ = 0
// (cpp/storage.cu:223-227) so the Adam bias moments never decay; SGD/Adagrad decay T by (1 − λ·lr).
// =============================================================================================
// the three bf16 pieces of x (gemm_split.hip split_pair: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round to nearest)
// stored where the projection products' kernels expect the element (k, n) of their B operand
__device__ __forceinline__ void store_plane_pieces(const PlaneTarget& t, int kw, int ne, float x) {
    if (!t.kind) return;
    const int k = t.transposed ? ne : kw, n = t.transposed ? kw : ne;
    size_t off;
    if (t.kind == 1) off = (static_cast<size_t>(k >> 5) * t.dim + n) * 64 + (k & 31) * 2;
    else off = ((static_cast<size_t>(k >> 4) * t.dim + (n >> 5)) * 64 + (n & 31) + 32 * ((k >> 3) & 1)) * 16 + (k & 7) * 2;
    const __bf16 h = static_cast<__bf16>(x);
    const float r = x - static_cast<float>(h);
    const __bf16 m = static_cast<__bf16>(r);
    const float s = r - static_cast<float>(m);
    const __bf16 l = static_cast<__bf16>(s);
    *reinterpret_cast<__bf16*>(t.planes + off) = h;
    *reinterpret_cast<__bf16*>(t.planes + off + t.plane_stride) = m;
    *reinterpret_cast<__bf16*>(t.planes + off + 2 * t.plane_stride) = l;
}

__device__ __forceinline__ void transform_update_element(const TransformUpdateArgs& a, bool is_bias, int i, float* P, float* G, float g, float p) {
    if (a.method == 0) {                                             // SGD  (updates.cu:24-34)
        const float dec = is_bias ? 1.f : static_cast<float>(1.0 - static_cast<double>(a.lambda) * static_cast<double>(a.lr));
        if (a.partial && !is_bias) *G = g;                           // (the slabs' sum: what launch_splitk_reduce would have left in gT)
        *P = p * dec + g * a.lr;
    } else if (a.method == 1) {                                      // Adagrad (updates_adagrad.cu:33-70)
        float* A = is_bias ? a.s0b + (i - a.nT) : a.s0T + i;
        const float acc = *A * 1.f + (g * g) * 1.f;
        *A = acc;
        g = g / sqrtf(acc + a.eps);
        *G = g;
        const float dec = is_bias ? 1.f : static_cast<float>(1.0 - static_cast<double>(a.lambda) * static_cast<double>(a.lr));
        *P = p * dec + g * a.lr;
    } else {                                                         // Adam (updates_adam.cu:46-105)
        float* M = is_bias ? a.s0b + (i - a.nT) : a.s0T + i;
        float* Vv = is_bias ? a.s1b + (i - a.nT) : a.s1T + i;
        if (!is_bias) g += (-a.lambda) * p;                          // apply_regularization: T only
        const float m = *M * (is_bias ? 1.f : a.s_m) + g * a.one_m_b1;
        const float v = *Vv * (is_bias ? 1.f : a.s_v) + (g * g) * a.one_m_b2;
        *M = m;
        *Vv = v;
        g = (m * a.bc) / (sqrtf(v) + a.eps);
        *G = g;
        *P = p * 1.f + g * a.lr;
    }
}

__global__ void transform_update_kernel(TransformUpdateArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nT + a.nb) return;
    const bool is_bias = i >= a.nT;
    float* P = is_bias ? a.b + (i - a.nT) : a.T + i;
    float* G = is_bias ? a.gb + (i - a.nT) : a.gT + i;
    float g;
    if (a.partial && !is_bias) {
        // splitk_reduce_kernel's sum (gather_gemm.hip): group j adds slabs j, j + 16, j + 32, ... in that order, then the sixteen
        // group sums are added in group order
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int z0 = 0; z0 < a.slabs; z0 += 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (z0 + j < a.slabs) acc[j] += a.partial[static_cast<size_t>(z0 + j) * a.slab_stride + i];
        }
        g = acc[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) g += acc[j];
    } else {
        g = *G;
    }
    float p = *P;
    transform_update_element(a, is_bias, i, P, G, g, p);
    if (!is_bias && (a.pt[0].kind | a.pt[1].kind)) {
        const float x = *P;                                          // (the value just stored: same thread)
        const int kw = i / a.de, ne = i - kw * a.de;
        store_plane_pieces(a.pt[0], kw, ne, x);
        store_plane_pieces(a.pt[1], kw, ne, x);
    }
}

void launch_transform_update(const TransformUpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(transform_update_kernel, dim3(ceil_div(a.nT + a.nb, 256)), dim3(256), 0, s, a);
}

}  // namespace cunvsm
