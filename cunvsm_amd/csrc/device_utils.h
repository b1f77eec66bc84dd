// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cunvsm {

template <int V>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&x)[V]) {
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) x[i] = p[i];
    }
}

template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&x)[V]) {
    if constexpr (V == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = x[i];
    }
}

// streaming variants (nt): data that is touched once per step and should not displace the gather sources in the caches
template <int V>
__device__ __forceinline__ void ldv_nt(const float* __restrict__ p, float (&x)[V]) {
    if constexpr (V == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        x[0] = t[0]; x[1] = t[1]; x[2] = t[2]; x[3] = t[3];
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) x[i] = __builtin_nontemporal_load(p + i);
    }
}
template <int V>
__device__ __forceinline__ void stv_nt(float* __restrict__ p, const float (&x)[V]) {
    if constexpr (V == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t; t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; t[3] = x[3];
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) __builtin_nontemporal_store(x[i], p + i);
    }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// Wave64 sum on the DPP cross-lane network (no LDS round trips): inclusive scan inside each 16-lane row
// (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 fold the four row totals into lane 63, whose value is
// returned to every lane through an SGPR. ~6 VALU issues instead of a 6-deep ds_bpermute dependency chain.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0x111, 0xf>(v);   // row_shr:1
    v += dpp_mov<0x112, 0xf>(v);   // row_shr:2
    v += dpp_mov<0x114, 0xf>(v);   // row_shr:4
    v += dpp_mov<0x118, 0xf>(v);   // row_shr:8
    v += dpp_mov<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v += dpp_mov<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
    // gfx950 has a native global_atomic_add_f64
    unsafeAtomicAdd(p, v);
}

// ---- hand-over of values between workgroups of ONE launch --------------------------------------------------------------
// The XCDs' L2s are not coherent with each other inside a kernel. A value that another workgroup of the same launch will read
// travels as agent-scope atomic stores / loads (write-through, cache-bypassing) — no release / acquire fence, which would
// write back and invalidate whole caches under the step's bandwidth-bound kernels. The writer waits for its stores
// (s_waitcnt vmcnt(0)) before it bumps the arrival counter that tells the others.
__device__ __forceinline__ void st_agent1(float* p, float x) {
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned int*>(const_cast<float*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent_f64(double* p, double x) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(x)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent_f64(const double* p) {
    return __longlong_as_double(static_cast<long long>(
        __hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<double*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}

// Ordered grid-wide sum: `total` workgroups each contribute a vector of n floats (val(i)); out(i, Σ) receives the sums.
// Replaces one fp64 atomicAdd per value and workgroup (column statistics of the batch-norm / bias gradient, the loss word):
// atomics add in arrival order, so the last bits differed from run to run, and a thousand workgroups adding to the same
// 513 addresses serialise in the L2's atomic units (a quarter of the loss kernel's time at batch 6 400). Here a workgroup
// stores its vector, bumps the arrival counter of its group of `fan` workgroups, and whoever arrives LAST — nobody waits —
// adds the group's vectors in member order (in double); the last group to finish adds the group sums in group order.
// Which workgroup that is varies; what it computes does not: results are the same bits every run. Counters return to
// zero for the next launch. part [total][n], part2 [ceil(total / fan)][n], arrive [ceil(total / fan) + 1] (zero before
// the first launch). Every thread of the workgroup must call it; `flag` is one int of LDS; nothing block-wide may follow.
// Memory order of the hand-over: the vectors travel as agent-scope write-through stores (st_agent1), every wave drains its
// stores (s_waitcnt vmcnt(0): on gfx9 / CDNA, stores count in vmcnt and a store has reached the agent-coherent L2 when the
// counter lets go), the workgroup meets, and only then one lane bumps the counter with a RELAXED agent-scope atomic; the last
// arriver reads the vectors with agent-scope loads. That is correct on gfx90a / gfx942 / gfx950 as written and is NOT what the
// HIP memory model promises in general (it would ask for a release on the bump and an acquire on the last arriver — an L2
// write-back and an invalidate per workgroup, which the write-through / cache-bypassing accesses make redundant here). Targets
// that count stores in a separate counter (vscnt: gfx10+) would let the last arriver read stale vectors: refused below.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "grid_sum_ordered orders its hand-over with s_waitcnt vmcnt(0) + relaxed agent-scope atomics: valid on gfx90a / gfx942 / gfx950 only"
#endif
// (vectors in flight per thread in the two sums of the last arrivers: a group of 32 members is then two dependent round trips
//  instead of four — the tail of a kernel that a whole step waits behind; the same additions in the same order)
#ifndef NVSM_GRID_SUM_IN_FLIGHT
#define NVSM_GRID_SUM_IN_FLIGHT 16
#endif
constexpr int kGridSumInFlight = NVSM_GRID_SUM_IN_FLIGHT;
template <int THREADS, class Val, class Out>
__device__ __forceinline__ void grid_sum_ordered(float* part, double* part2, int* arrive, int fan, int n, int me, int total,
                                                 Val val, Out out, int* flag) {
    const int tid = threadIdx.x;
    const int T = THREADS > 0 ? THREADS : static_cast<int>(blockDim.x);      // (THREADS = 0: the block size is a run-time value)
    if (total == 1) {                  // a single workgroup: nothing to hand over
        for (int i = tid; i < n; i += T) out(i, static_cast<double>(val(i)));
        return;
    }
    for (int i = tid; i < n; i += T) st_agent1(part + static_cast<size_t>(me) * n + i, val(i));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int grp = me / fan, ngroups = (total + fan - 1) / fan;
    const int members = min(fan, total - grp * fan);
    if (tid == 0) {
        int last = 0;
        if (__hip_atomic_fetch_add(arrive + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(arrive + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = 1;
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const float* gp = part + static_cast<size_t>(grp) * fan * n;
    for (int i = tid; i < n; i += T) {
        double s = 0.0;
        for (int j0 = 0; j0 < members; j0 += kGridSumInFlight) {        // members in flight, added in member order
            float v[kGridSumInFlight];
#pragma unroll
            for (int u = 0; u < kGridSumInFlight; ++u) v[u] = ld_agent1(gp + static_cast<size_t>(min(j0 + u, members - 1)) * n + i);
#pragma unroll
            for (int u = 0; u < kGridSumInFlight; ++u) s += (j0 + u < members) ? static_cast<double>(v[u]) : 0.0;
        }
        if (ngroups == 1) out(i, s); else st_agent_f64(part2 + static_cast<size_t>(grp) * n + i, s);
    }
    if (ngroups == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int last = 0;
        if (__hip_atomic_fetch_add(arrive + ngroups, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) {
            __hip_atomic_store(arrive + ngroups, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = 1;
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    for (int i = tid; i < n; i += T) {
        double s = 0.0;
        for (int g0 = 0; g0 < ngroups; g0 += kGridSumInFlight) {
            double v[kGridSumInFlight];
#pragma unroll
            for (int u = 0; u < kGridSumInFlight; ++u) v[u] = ld_agent_f64(part2 + static_cast<size_t>(min(g0 + u, ngroups - 1)) * n + i);
#pragma unroll
            for (int u = 0; u < kGridSumInFlight; ++u) s += (g0 + u < ngroups) ? v[u] : 0.0;
        }
        out(i, s);
    }
}

inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

// memory-bound launches: enough blocks to fill 256 CUs x 8, grid-stride the rest
inline int stream_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return static_cast<int>(g);
}

}  // namespace cunvsm
