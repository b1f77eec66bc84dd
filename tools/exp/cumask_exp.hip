// Is a CU mask honoured here? A compute-bound kernel on a stream confined to 1 of every 8 CUs should take ~8x as long.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(float* out, int n) {
    float a = threadIdx.x * 1e-3f;
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.f) out[0] = a;
}
int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* d; CHECK(hipMalloc(&d, 4));
    for (int eighths : {8, 4, 1}) {
        hipStream_t s;
        if (eighths == 8) CHECK(hipStreamCreate(&s));
        else {
            std::vector<uint32_t> m((p.multiProcessorCount + 31) / 32, 0u);
            for (int i = 0; i < p.multiProcessorCount; ++i) if ((i & 7) < eighths) m[i >> 5] |= 1u << (i & 31);
            CHECK(hipExtStreamCreateWithCUMask(&s, static_cast<uint32_t>(m.size()), m.data()));
        }
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(spin, dim3(256 * 32), dim3(256), 0, s, d, 20000);
        CHECK(hipEventRecord(a, s));
        hipLaunchKernelGGL(spin, dim3(256 * 32), dim3(256), 0, s, d, 20000);
        CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        std::printf("%d of every 8 CUs (%d CUs in all): %.3f ms\n", eighths, p.multiProcessorCount, ms);
    }
    return 0;
}
