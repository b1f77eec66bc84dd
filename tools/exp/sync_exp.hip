// Experiment (not product code): what does an edge between two streams cost, by primitive? A chain of spin kernels that
// alternates between two streams, every kernel waiting for the one before it on the other stream:
//   events (default flags) / hipEventDisableTiming / + hipEventDisableSystemFence : hipEventRecord + hipStreamWaitEvent
//   riding      : the event rides on the producer's launch as its stop event (hipExtLaunchKernelGGL) + hipStreamWaitEvent
//   stream memop: hipStreamWriteValue32 behind the producer + hipStreamWaitValue32 (>=) in front of the consumer, on
//                 signal memory (hipExtMallocWithFlags(hipMallocSignalMemory)), a counter that only grows
// against the same chain on ONE stream. Prints microseconds per hop beyond the kernel's own duration.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/sync_exp.hip -o tools/exp/sync_exp.out && tools/exp/sync_exp.out [kernel_us]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (ticks < 0) *sink = 1;
}

int main(int argc, char** argv) {
    const double kernel_us = argc > 1 ? std::atof(argv[1]) : 5.0;
    const long long t = static_cast<long long>(kernel_us * 100);
    const int hops = 2000;
    hipStream_t s[2];
    int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, lo));
    int* sink; CHECK(hipMalloc(&sink, 4));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    auto report = [&](const char* name, double total_us) {
        std::printf("%-58s %6.2f us per hop beyond the kernel\n", name, total_us / hops - kernel_us);
    };
    auto sync = [&] { CHECK(hipStreamSynchronize(s[0])); CHECK(hipStreamSynchronize(s[1])); };

    {   // one stream
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[0], t, sink);
        sync();
        auto t0 = now();
        for (int i = 0; i < hops; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[0], t, sink);
        sync();
        report("one stream (no edge)", us(t0, now()));
    }
    for (int mode = 0; mode < 3; ++mode) {      // plain events
        const unsigned flags = mode == 0 ? hipEventDefault : (mode == 1 ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence));
        std::vector<hipEvent_t> ev(8);
        for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, flags));
        auto run = [&](int n) {
            for (int i = 0; i < n; ++i) {
                hipStream_t st = s[i & 1];
                if (i > 0) CHECK(hipStreamWaitEvent(st, ev[(i - 1) & 7], 0));
                hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, t, sink);
                CHECK(hipEventRecord(ev[i & 7], st));
            }
        };
        run(50); sync();
        auto t0 = now();
        run(hops); sync();
        report(mode == 0 ? "events, default flags" : (mode == 1 ? "events, hipEventDisableTiming" : "events, DisableTiming | DisableSystemFence"), us(t0, now()));
        for (auto& e : ev) CHECK(hipEventDestroy(e));
    }
    {   // riding stop events
        std::vector<hipEvent_t> ev(8);
        for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
        auto run = [&](int n) {
            for (int i = 0; i < n; ++i) {
                hipStream_t st = s[i & 1];
                if (i > 0) CHECK(hipStreamWaitEvent(st, ev[(i - 1) & 7], 0));
                hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, ev[i & 7], 0, t, sink);
            }
        };
        run(50); sync();
        auto t0 = now();
        run(hops); sync();
        report("stop event riding on the launch (hipExtLaunchKernelGGL)", us(t0, now()));
        for (auto& e : ev) CHECK(hipEventDestroy(e));
    }
    {   // stream memory operations on signal memory
        uint32_t* flag[2];
        bool ok = true;
        for (int k = 0; k < 2; ++k)
            if (hipExtMallocWithFlags(reinterpret_cast<void**>(&flag[k]), 8, hipMallocSignalMemory) != hipSuccess) { ok = false; (void)hipGetLastError(); }
        if (ok) {
            for (int k = 0; k < 2; ++k) CHECK(hipMemset(flag[k], 0, 8));
            uint32_t count = 0;
            auto run = [&](int n) {
                for (int i = 0; i < n; ++i) {
                    const int k = i & 1;
                    if (count > 0) CHECK(hipStreamWaitValue32(s[k], flag[k ^ 1], count, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[k], t, sink);
                    ++count;
                    CHECK(hipStreamWriteValue32(s[k], flag[k], count, 0));
                }
            };
            run(50); sync();
            auto t0 = now();
            run(hops); sync();
            report("hipStreamWriteValue32 / hipStreamWaitValue32 (signal memory)", us(t0, now()));
        } else std::printf("stream memory operations: hipMallocSignalMemory refused\n");
    }
    // The two halves of an edge, each alone on ONE stream's chain: what a record nobody waits for costs the stream that makes it,
    // and what a wait for something long complete costs the stream that waits.
    {
        hipEvent_t ev, done; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
        CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming | hipEventDisableSystemFence));
        CHECK(hipEventRecord(done, s[1])); CHECK(hipStreamSynchronize(s[1]));
        uint32_t* flag = nullptr;
        const bool memops = hipExtMallocWithFlags(reinterpret_cast<void**>(&flag), 8, hipMallocSignalMemory) == hipSuccess;
        if (memops) CHECK(hipMemset(flag, 0, 8));
        uint32_t count = 0;
        for (int mode = 0; mode < 5; ++mode) {
            if (mode >= 3 && !memops) break;
            auto run = [&](int n) {
                for (int i = 0; i < n; ++i) {
                    if (mode == 2) CHECK(hipStreamWaitEvent(s[0], done, 0));
                    if (mode == 4) CHECK(hipStreamWaitValue32(s[0], flag, 0, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[0], nullptr, ev, 0, t, sink);
                    else hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[0], t, sink);
                    if (mode == 0) CHECK(hipEventRecord(ev, s[0]));
                    if (mode == 3) CHECK(hipStreamWriteValue32(s[0], flag, ++count, 0));
                }
            };
            run(50); sync();
            auto t0 = now();
            run(hops); sync();
            const char* names[5] = {"one stream + hipEventRecord behind every kernel", "one stream, every kernel carries a stop event",
                                    "one stream + a wait for a long-complete event before every kernel",
                                    "one stream + hipStreamWriteValue32 behind every kernel", "one stream + a satisfied hipStreamWaitValue32 before every kernel"};
            report(names[mode], us(t0, now()));
        }
    }
    return 0;
}
