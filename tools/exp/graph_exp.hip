// Experiment (not product code): does a captured hipGraph run a small multi-stream step faster than the same launches issued
// eagerly? The batch-4 096 / 6 400 steps are ~25 launches on four streams tied together by ~6 events; their kernels sum to
// ~135 / ~225 us and the steps take 164 / 290 — the rest is gaps at the dependent launches. This program issues a step of the
// same SHAPE (spin kernels of fixed duration: a main chain of eight, a side chain of six forked at the start and joined
// before the sixth main kernel, a second side chain of three forked behind the fourth main kernel and joined by the NEXT
// step's second kernel) three ways and reports the time per step:
//   eager    — hipLaunchKernelGGL + hipEventRecord / hipStreamWaitEvent, the host running ahead (as nvsm_step does)
//   graph    — one step captured by stream capture (the tail joined at the end of the step: a graph cannot leave work
//              running into the next launch), replayed with hipGraphLaunch
//   graph x4 — four steps captured as ONE graph (the tail of step k joined by step k + 1 inside the graph)
//   hipcc -O3 --offload-arch=gfx950 tools/exp/graph_exp.hip -o tools/exp/graph_exp.out && tools/exp/graph_exp.out [kernel_us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void spin(long long ticks, int* sink) {      // 100 MHz wall clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (ticks < 0) *sink = 1;
}

struct Streams { hipStream_t main, a, b; hipEvent_t fork_a, join_a, fork_b, join_b; };
static int* g_sink;

// one step; `join_tail_here`: the second side chain is joined at the end of this step instead of by the next one
static void step(const Streams& s, long long t, bool first, bool join_tail_here) {
    auto k = [&](hipStream_t st, int n = 64) { hipLaunchKernelGGL(spin, dim3(n), dim3(256), 0, st, t, g_sink); };
    k(s.main);
    CHECK(hipEventRecord(s.fork_a, s.main)); CHECK(hipStreamWaitEvent(s.a, s.fork_a, 0));
    for (int i = 0; i < 6; ++i) k(s.a, 16);
    CHECK(hipEventRecord(s.join_a, s.a));
    if (!first && !join_tail_here) CHECK(hipStreamWaitEvent(s.main, s.join_b, 0));      // the previous step's tail
    k(s.main); k(s.main); k(s.main);
    CHECK(hipEventRecord(s.fork_b, s.main)); CHECK(hipStreamWaitEvent(s.b, s.fork_b, 0));
    for (int i = 0; i < 3; ++i) k(s.b, 16);
    CHECK(hipEventRecord(s.join_b, s.b));
    k(s.main);
    CHECK(hipStreamWaitEvent(s.main, s.join_a, 0));
    k(s.main); k(s.main); k(s.main);
    if (join_tail_here) CHECK(hipStreamWaitEvent(s.main, s.join_b, 0));
}

int main(int argc, char** argv) {
    const double kernel_us = argc > 1 ? std::atof(argv[1]) : 8.0;
    const long long t = static_cast<long long>(kernel_us * 100);
    Streams s;
    int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithPriority(&s.main, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&s.a, hipStreamNonBlocking, lo));
    CHECK(hipStreamCreateWithPriority(&s.b, hipStreamNonBlocking, lo));
    for (hipEvent_t* e : {&s.fork_a, &s.join_a, &s.fork_b, &s.join_b}) CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    CHECK(hipMalloc(&g_sink, 4));
    const int steps = 400;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    auto sync_all = [&] { CHECK(hipStreamSynchronize(s.main)); CHECK(hipStreamSynchronize(s.a)); CHECK(hipStreamSynchronize(s.b)); };

    // eager
    for (int i = 0; i < 20; ++i) step(s, t, i == 0, false);
    sync_all();
    auto t0 = now();
    for (int i = 0; i < steps; ++i) step(s, t, false, false);
    auto t_host = now();
    sync_all();
    auto t1 = now();
    std::printf("kernels of %.1f us; main chain 8 kernels = %.0f us of work per step\n", kernel_us, 8 * kernel_us);
    std::printf("eager            : %7.1f us per step (host queued a step in %.1f us)\n", us(t0, t1) / steps, us(t0, t_host) / steps);

    for (int unroll : {1, 4}) {
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s.main, hipStreamCaptureModeRelaxed));
        for (int u = 0; u < unroll; ++u) step(s, t, u == 0, u == unroll - 1);
        CHECK(hipStreamEndCapture(s.main, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CHECK(hipGraphLaunch(exec, s.main));
        CHECK(hipStreamSynchronize(s.main));
        t0 = now();
        for (int i = 0; i < steps / unroll; ++i) CHECK(hipGraphLaunch(exec, s.main));
        t_host = now();
        CHECK(hipStreamSynchronize(s.main));
        t1 = now();
        std::printf("graph of %d step%s : %7.1f us per step (host launched a step in %.1f us)\n", unroll, unroll > 1 ? "s" : " ",
                    us(t0, t1) / steps, us(t0, t_host) / steps);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    }
    {   // everything on ONE stream (17 kernels in a row), eager and as a graph: no edges between streams at all
        Streams one = s; one.a = one.b = s.main;
        auto run = [&](const Streams& st) {
            auto k = [&](int n = 64) { hipLaunchKernelGGL(spin, dim3(n), dim3(256), 0, st.main, t, g_sink); };
            k(); for (int i = 0; i < 6; ++i) k(16); k(); k(); k(); for (int i = 0; i < 3; ++i) k(16); k(); k(); k(); k();
        };
        for (int i = 0; i < 5; ++i) run(one);
        CHECK(hipStreamSynchronize(s.main));
        t0 = now();
        for (int i = 0; i < steps; ++i) run(one);
        CHECK(hipStreamSynchronize(s.main));
        t1 = now();
        std::printf("one stream eager : %7.1f us per step (17 kernels in a row = %.0f us of work)\n", us(t0, t1) / steps, 17 * kernel_us);
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s.main, hipStreamCaptureModeRelaxed));
        run(one);
        CHECK(hipStreamEndCapture(s.main, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CHECK(hipGraphLaunch(exec, s.main));
        CHECK(hipStreamSynchronize(s.main));
        t0 = now();
        for (int i = 0; i < steps; ++i) CHECK(hipGraphLaunch(exec, s.main));
        CHECK(hipStreamSynchronize(s.main));
        t1 = now();
        std::printf("one stream graph : %7.1f us per step\n", us(t0, t1) / steps);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    }
    // the same main chain alone on one stream: what the dependent launches cost without any event
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 8; ++j) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s.main, t, g_sink);
    CHECK(hipStreamSynchronize(s.main));
    t0 = now();
    for (int i = 0; i < steps; ++i) for (int j = 0; j < 8; ++j) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s.main, t, g_sink);
    CHECK(hipStreamSynchronize(s.main));
    t1 = now();
    std::printf("main chain alone : %7.1f us per step (eight back-to-back launches on one stream, no events)\n", us(t0, t1) / steps);
    return 0;
}
