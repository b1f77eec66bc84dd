// Experiment (not product code): what does a wait that is ALREADY SATISFIED when the stream reaches it cost that stream?
// (sync_exp.hip prices a hop whose consumer really waits.) The step's main stream carries four such waits at the small batches —
// the projection update, the previous documents pass, the words CSR, the hoisted decay: all long finished in steady state — and
// the timelines show 6-10 us of idle stream at each.
// Stream A runs a chain of 5 us spin kernels; between every two of them sits one of:
//   nothing | hipStreamWaitEvent on an event recorded on stream B one turn earlier (B's kernel is 1 us: long finished when A gets there,
//   but NOT finished when the host enqueues the wait: the host runs a whole chain ahead) | the same with the event riding on B's launch |
//   hipStreamWaitValue32 (>=) on signal memory written by hipStreamWriteValue32 on B | a completion event riding on A's own kernel
//   (what a kernel that others wait for costs its own stream).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/wait_exp.hip -o tools/exp/wait_exp.out && tools/exp/wait_exp.out
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

int main() {
    const double kernel_us = 5.0;
    const long long t = static_cast<long long>(kernel_us * 100), tb = 100;      // B's kernels: 1 us
    const int n = 1000;
    hipStream_t a, b;
    int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo));
    int* sink; CHECK(hipMalloc(&sink, 4));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    auto sync = [&] { CHECK(hipStreamSynchronize(a)); CHECK(hipStreamSynchronize(b)); };
    // A is held for 12 ms and B for 1 ms: the host queues both chains behind the gates (no event has fired when its wait is enqueued:
    // the runtime cannot elide the packet), B's chain is through long before A's gate opens
    auto gate = [&] { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 1200000LL, sink); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 100000LL, sink); };
    auto report = [&](const char* name, double total) { std::printf("%-86s %6.2f us per turn beyond the kernel\n", name, (total - 12000.0) / n - kernel_us); };
    const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
    std::vector<hipEvent_t> ev(16);
    for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, flags));
    unsigned* sig = nullptr;
    CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 8, hipMallocSignalMemory));
    CHECK(hipMemset(sig, 0, 8));
    CHECK(hipDeviceSynchronize());
    unsigned counter = 0;

    for (int mode = 0; mode < 6; ++mode) {
        auto chain = [&](int turns) {
            // B's work of turn i is queued (and, on the GPU, long done) before A's wait of turn i + 8 refers to it
            for (int i = 0; i < turns; ++i) {
                if (mode == 1 || mode == 3) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, tb, sink); CHECK(hipEventRecord(ev[i & 15], b)); }
                if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, nullptr, ev[i & 15], 0, tb, sink);
                if (mode == 4) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, tb, sink); CHECK(hipStreamWriteValue32(b, sig, ++counter, 0)); }
                if (i >= 8) {
                    if (mode == 1 || mode == 2) CHECK(hipStreamWaitEvent(a, ev[(i - 8) & 15], 0));
                    if (mode == 3) { CHECK(hipStreamWaitEvent(a, ev[(i - 8) & 15], 0)); CHECK(hipStreamWaitEvent(a, ev[(i - 7) & 15], 0)); }      // two waits back to back
                    if (mode == 4) CHECK(hipStreamWaitValue32(a, sig, counter - 8, hipStreamWaitValueGte, 0xffffffffu));
                }
                if (mode == 5) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, nullptr, ev[i & 15], 0, t, sink);
                else hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, t, sink);
            }
        };
        gate(); chain(50); sync();
        auto t0 = now();
        gate(); chain(n); sync();
        const char* names[] = {"nothing between the kernels", "hipStreamWaitEvent, event recorded on B long before (plain record)",
                               "hipStreamWaitEvent, event riding on B's launch", "TWO such waits back to back",
                               "hipStreamWaitValue32 (>=) on signal memory written by hipStreamWriteValue32 on B",
                               "no wait, but every kernel of A carries a completion event (hipExtLaunchKernelGGL)"};
        report(names[mode], us(t0, now()));
    }
    return 0;
}
