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

inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

// memory-bound launches: enough blocks to fill 256 CUs x 8, grid-stride the rest
inline int stream_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return static_cast<int>(g);
}

}  // namespace cunvsm
