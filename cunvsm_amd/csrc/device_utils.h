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

// all 64 lanes receive the sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
