// Experiment (not product code): can the re-gather of a table pass be served by the XCDs' L2s instead of the fabric?
//
// The documents pass of the NVSM step sums, for each of 100 k table rows, the ~8.7 rows of `proj` [51 200][256] its entries
// point at: 891 MB of 1 KB gathers out of a 52 MB source that no 4 MB L2 holds — they cross the fabric (Infinity Cache hits or
// not), and the fabric is what the step's back half is bound by. Idea under test: store the source
// SLICED by columns, [256 / COLS][B][COLS] fp32, so that one slice is a contiguous 3.3 MB (COLS = 16) or 6.5 MB (COLS = 32),
// and let XCD x (workgroups with blockIdx % 8 == x: the dispatcher deals workgroups round-robin over the XCDs) own the columns
// [32 x, 32 x + 32) of EVERY table row: its gathers then touch its own slice(s) only, which its L2 can hold.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/slice_exp.hip -o tools/exp/slice_exp.out && tools/exp/slice_exp.out [variant]
//   variant 0 = whole rows (one wave per table row, 1 KB gathers: today's shape), 1 = COLS 16 (two phases per XCD), 2 = COLS 32
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kDim = 256;

__device__ __forceinline__ f32x4 ld_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void st_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

// variant 0: one wave per table row, lane l owns columns 4 l .. 4 l + 3; four gradient rows in flight
__global__ __launch_bounds__(256) void rows_kernel(const float* __restrict__ S, const int* __restrict__ row_begin, const int* __restrict__ eb,
                                                   const float* __restrict__ coef, float* __restrict__ P, float* __restrict__ M, int rows) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const int b0 = row_begin[r], b1 = row_begin[r + 1];
        f32x4 g = {0, 0, 0, 0};
        for (int e = b0; e < b1; e += 4) {
            f32x4 x[4]; float c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = min(e + u, b1 - 1);
                c[u] = (e + u < b1) ? coef[ee] : 0.f;
                x[u] = *reinterpret_cast<const f32x4*>(S + static_cast<size_t>(eb[ee]) * kDim + 4 * lane);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) g += c[u] * x[u];
        }
        float* p = P + static_cast<size_t>(r) * kDim + 4 * lane;
        float* m = M + static_cast<size_t>(r) * kDim + 4 * lane;
        const f32x4 mm = 0.9f * ld_nt(m) + 0.1f * g;
        st_nt(m, mm);
        st_nt(p, ld_nt(p) - 0.001f * mm);
    }
}

// variants 1, 2: XCD x = blockIdx % 8 owns columns [32 x, 32 x + 32) of every table row, in 32 / COLS phases of COLS columns;
// COLS / 4 lanes per table row, 256 / (COLS / 4) rows per workgroup; blockIdx = (phase * groups + group) * 8 + xcd
template <int COLS>
__global__ __launch_bounds__(256) void slices_kernel(const float* __restrict__ S, int B, const int* __restrict__ row_begin, const int* __restrict__ eb,
                                                     const float* __restrict__ coef, float* __restrict__ P, float* __restrict__ M, int rows, int groups) {
    constexpr int LPR = COLS / 4, RPW = 256 / LPR, PHASES = 32 / COLS;
    const int xcd = blockIdx.x & 7;
    const int pg = blockIdx.x >> 3;
    const int phase = pg / groups, group = pg - phase * groups;
    const int slice = xcd * PHASES + phase;
    const int sub = threadIdx.x % LPR;
    const float* Ss = S + static_cast<size_t>(slice) * B * COLS + 4 * sub;
    const int r = group * RPW + threadIdx.x / LPR;
    if (r >= rows) return;
    const int b0 = row_begin[r], b1 = row_begin[r + 1];
    f32x4 g = {0, 0, 0, 0};
    for (int e = b0; e < b1; e += 4) {
        f32x4 x[4]; float c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ee = min(e + u, b1 - 1);
            c[u] = (e + u < b1) ? coef[ee] : 0.f;
            x[u] = *reinterpret_cast<const f32x4*>(Ss + static_cast<size_t>(eb[ee]) * COLS);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) g += c[u] * x[u];
    }
    float* p = P + static_cast<size_t>(r) * kDim + slice * COLS + 4 * sub;
    float* m = M + static_cast<size_t>(r) * kDim + slice * COLS + 4 * sub;
    const f32x4 mm = 0.9f * ld_nt(m) + 0.1f * g;
    st_nt(m, mm);
    st_nt(p, ld_nt(p) - 0.001f * mm);
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? std::atoi(argv[1]) : -1;
    const int B = argc > 2 ? std::atoi(argv[2]) : 51200, rows = argc > 3 ? std::atoi(argv[3]) : 100000, per = 17;
    const int reps = 20;
    const size_t N = static_cast<size_t>(B) * per;
    std::mt19937 rng(7);
    std::vector<int> doc(N), order(N);
    for (auto& d : doc) d = static_cast<int>(rng() % rows);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return doc[a] < doc[b]; });
    std::vector<int> row_begin(rows + 1, 0), eb(N);
    std::vector<float> coef(N);
    for (size_t i = 0; i < N; ++i) { row_begin[doc[order[i]] + 1]++; eb[i] = order[i] / per; coef[i] = 0.25f + (order[i] % 7) * 0.125f; }
    for (int r = 0; r < rows; ++r) row_begin[r + 1] += row_begin[r];
    std::vector<float> S0(static_cast<size_t>(B) * kDim);
    for (auto& v : S0) v = static_cast<float>(static_cast<int>(rng() % 2001) - 1000) * 1e-3f;
    auto sliced = [&](int cols) {
        std::vector<float> s(S0.size());
        for (int b = 0; b < B; ++b) for (int c = 0; c < kDim; ++c) s[(static_cast<size_t>(c / cols) * B + b) * cols + c % cols] = S0[static_cast<size_t>(b) * kDim + c];
        return s;
    };
    float *dS[3], *dP, *dM; int *dRb, *dEb; float* dC;
    const std::vector<float> s16 = sliced(16), s32 = sliced(32);
    const std::vector<float>* hs[3] = {&S0, &s16, &s32};
    for (int v = 0; v < 3; ++v) { CHECK(hipMalloc(&dS[v], S0.size() * 4)); CHECK(hipMemcpy(dS[v], hs[v]->data(), S0.size() * 4, hipMemcpyHostToDevice)); }
    const size_t tab = static_cast<size_t>(rows) * kDim;
    CHECK(hipMalloc(&dP, tab * 4)); CHECK(hipMalloc(&dM, tab * 4));
    CHECK(hipMalloc(&dRb, (rows + 1) * 4)); CHECK(hipMalloc(&dEb, N * 4)); CHECK(hipMalloc(&dC, N * 4));
    CHECK(hipMemcpy(dRb, row_begin.data(), (rows + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dEb, eb.data(), N * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dC, coef.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<float> ref, got(tab);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int v = 0; v < 3; ++v) {
        if (variant >= 0 && v != variant) continue;
        auto launch = [&] {
            if (v == 0) hipLaunchKernelGGL(rows_kernel, dim3(256 * 8), dim3(256), 0, 0, dS[0], dRb, dEb, dC, dP, dM, rows);
            else if (v == 1) { const int groups = (rows + 63) / 64; hipLaunchKernelGGL(slices_kernel<16>, dim3(2 * groups * 8), dim3(256), 0, 0, dS[1], B, dRb, dEb, dC, dP, dM, rows, groups); }
            else { const int groups = (rows + 31) / 32; hipLaunchKernelGGL(slices_kernel<32>, dim3(groups * 8), dim3(256), 0, 0, dS[2], B, dRb, dEb, dC, dP, dM, rows, groups); }
        };
        CHECK(hipMemset(dP, 0, tab * 4)); CHECK(hipMemset(dM, 0, tab * 4));
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), dP, tab * 4, hipMemcpyDeviceToHost));
        if (ref.empty()) ref = got;
        const bool same = std::memcmp(ref.data(), got.data(), tab * 4) == 0;
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double gather_mb = N * 1024.0 / 1e6, state_mb = tab * 4.0 * 4 / 1e6;
        std::printf("variant %d (%s): %.1f us per launch; gathers %.0f MB + table state %.0f MB = %.2f TB/s algorithmic; result %s\n", v,
                    v == 0 ? "whole rows, 1 KB gathers" : (v == 1 ? "16-column slices, XCD-owned, 2 phases" : "32-column slices, XCD-owned"),
                    ms * 1e3 / reps, gather_mb, state_mb, (gather_mb + state_mb) / (ms * 1e3 / reps), same ? "identical to variant 0" : "DIFFERENT");
    }
    return 0;
}
