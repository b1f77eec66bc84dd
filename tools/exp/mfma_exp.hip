// Experiment harness (not product code): what rate does v_mfma_f32_16x16x4_f32 / 32x32x2 sustain in this kernel shape?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int TM, int TN>
__global__ __launch_bounds__(256, 1) void pure16(float* out, const float* in, int iters) {
    f32x4 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    float a[TM], b[TN];
    for (int i = 0; i < TM; ++i) a[i] = in[threadIdx.x + 64 * i];
    for (int j = 0; j < TN; ++j) b[j] = in[threadIdx.x + 64 * (TM + j)];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int TM, int TN>
__global__ __launch_bounds__(256, 1) void lds16(float* out, const float* in, int iters) {
    __shared__ float lds[16 * TM * 33 + 32 * (64 * TN + 4)];
    float* As = lds; float* Bs = lds + 16 * TM * 33;
    for (int i = threadIdx.x; i < 16 * TM * 33 + 32 * (64 * TN + 4); i += 256) lds[i] = in[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    f32x4 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 32; ks += 4) {
            const int k = ks + lg;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(i * 16 + li) * 33 + k];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[k * (64 * TN + 4) + (wid * TN + j) * 16 + li];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int T>
__global__ __launch_bounds__(256, 1) void pure32(float* out, const float* in, int iters) {
    f32x16 acc[T][T];
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    float a[T], b[T];
    for (int i = 0; i < T; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * (T + i)]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out, *in; CK(hipMalloc(&out, 1 << 24)); CK(hipMalloc(&in, 1 << 20));
    { float* h = (float*)malloc(1 << 20); for (int i = 0; i < (1 << 18); ++i) h[i] = ((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f; CK(hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* nm, double flops, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); }
        printf("%-40s %.1f us  %.1f TF/s\n", nm, best * 1e3, flops / (best * 1e-3) / 1e12);
    };
    const int it16 = 640;   // 52 mfma * 640 = 33280 mfma per wave
    timeit("pure mfma16x16x4 13x4, 256 blk x 4 waves", 256.0 * 4 * 52 * it16 * 2048, [&] { pure16<13, 4><<<256, 256>>>(out, in, it16); });
    timeit("pure mfma16x16x4 7x4, 512 blk", 512.0 * 4 * 28 * it16 * 2048, [&] { pure16<7, 4><<<512, 256>>>(out, in, it16); });
    timeit("lds+mfma16 13x4 (8 steps/iter), 256 blk", 256.0 * 4 * 52 * 8 * 80 * 2048, [&] { lds16<13, 4><<<256, 256>>>(out, in, 80); });
    timeit("lds+mfma16 7x4, 512 blk", 512.0 * 4 * 28 * 8 * 80 * 2048, [&] { lds16<7, 4><<<512, 256>>>(out, in, 80); });
    timeit("pure mfma32x32x2 2x2, 768 blk", 768.0 * 4 * 4 * 4000 * 4096, [&] { pure32<2><<<768, 256>>>(out, in, 4000); });
    timeit("pure mfma32x32x2 2x2, 256 blk", 256.0 * 4 * 4 * 8000 * 4096, [&] { pure32<2><<<256, 256>>>(out, in, 8000); });
    return 0;
}
