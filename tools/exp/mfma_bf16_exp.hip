// Experiment harness (not product code): cycles per v_mfma_f32_16x16x32_bf16 as a function of the number of independent
// accumulators a wave rotates through (the dependency distance) and of the waves per SIMD. hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int D>
__global__ __launch_bounds__(512) void chain(float* out, const unsigned* in, int iters, unsigned long long* cyc) {
    f32x4 acc[D];
    for (int i = 0; i < D; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 ua = *reinterpret_cast<const u4*>(in + 4 * threadIdx.x), ub = *reinterpret_cast<const u4*>(in + 4 * threadIdx.x + 4096);
        a = __builtin_bit_cast(bf16x8, ua); b = __builtin_bit_cast(bf16x8, ub);
    }
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / D; ++r)
#pragma unroll
            for (int i = 0; i < D; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < D; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int D>
void run(int waves, float* out, unsigned* in, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(chain<D>, dim3(256), dim3(64 * waves), 0, 0, out, in, iters, cyc);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double n = double(iters) * (48 / D) * D;
    // waves per SIMD = waves / 4: the SIMD's MFMA count in the interval is n x (waves / 4)
    printf("accumulators in rotation %d, %d waves per SIMD: %.1f cycles per MFMA and wave, %.1f per MFMA and SIMD\n", D, waves / 4,
           h[0] / n, h[0] / (n * (waves / 4)));
}

int main() {
    float* out; unsigned* in; unsigned long long* cyc;
    CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&in, 1 << 16)); CK(hipMalloc(&cyc, 64));
    CK(hipMemset(in, 0x3c, 1 << 16));
    for (int waves : {4, 8}) {
        run<1>(waves, out, in, cyc); run<2>(waves, out, in, cyc); run<3>(waves, out, in, cyc); run<4>(waves, out, in, cyc); run<6>(waves, out, in, cyc); run<8>(waves, out, in, cyc);
    }
    return 0;
}
