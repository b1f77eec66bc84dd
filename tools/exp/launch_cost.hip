// Experiment (not product code): host cost of queueing a kernel by API path — hipLaunchKernelGGL (<<<>>>: push / pop call configuration +
// hipLaunchKernel), hipLaunchKernel with a prepared argument array, hipModuleLaunchKernel on a hipFunction_t from hipGetFuncBySymbol,
// hipExtLaunchKernelGGL with a stop event. The stream is kept from filling up by a synchronise every 256 launches (not timed).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k(int* p, int a, float b, const int* q, long c) { if (p && threadIdx.x == 9999) p[0] = a + (int)b + (q ? q[0] : 0) + (int)c; }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* p = nullptr; hipMalloc(&p, 64);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int N = 20000;
    auto run = [&](const char* name, auto&& f) {
        for (int i = 0; i < 512; ++i) f();
        hipStreamSynchronize(s);
        double total = 0;
        for (int r = 0; r < N / 256; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 256; ++i) f();
            total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            hipStreamSynchronize(s);
        }
        std::printf("%-60s %.2f us per call\n", name, total / (N / 256 * 256));
    };
    int a = 1; float b = 2.f; const int* q = nullptr; long c = 3;
    run("hipLaunchKernelGGL", [&] { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, p, a, b, q, c); });
    void* args[] = {&p, &a, &b, &q, &c};
    run("hipLaunchKernel (argument pointers)", [&] { hipLaunchKernel(reinterpret_cast<const void*>(k), dim3(1), dim3(64), args, 0, s); });
    hipFunction_t fn = nullptr;
    if (hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(k)) == hipSuccess && fn) {
        run("hipModuleLaunchKernel (hipFunction_t, kernelParams)", [&] { hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr); });
        struct { int* p; int a; float b; const int* q; long c; } pk{p, a, b, q, c};
        size_t sz = sizeof(pk);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        run("hipModuleLaunchKernel (packed buffer)", [&] { hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, s, nullptr, cfg); });
    } else std::printf("hipGetFuncBySymbol unavailable\n");
    run("hipExtLaunchKernelGGL + stop event", [&] { hipExtLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, nullptr, ev, 0, p, a, b, q, c); });
    run("hipEventRecord", [&] { hipEventRecord(ev, s); });
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEventRecord(ev, s2); hipStreamSynchronize(s2);
    run("hipStreamWaitEvent (fired event)", [&] { hipStreamWaitEvent(s, ev, 0); });
    run("hipGetDevice", [&] { int d; hipGetDevice(&d); });
    run("hipSetDevice", [&] { hipSetDevice(0); });
    run("hipGetLastError", [&] { (void)hipGetLastError(); });
    return 0;
}
