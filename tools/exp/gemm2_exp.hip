// Experiment harness (not product code): LDS-stationary projection GEMM (gemm_tstat.hip) vs the 128 x 128 tiled kernel.

#include "../../cunvsm_amd/csrc/gather_gemm.hip"
#include "../../cunvsm_amd/csrc/gemm_panel.hip"
#include "../../cunvsm_amd/csrc/gemm_tstat.hip"
#include "../../cunvsm_amd/csrc/loss_bn.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace cunvsm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void fill_rand(float* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        p[i] = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.f * scale;
    }
}
int main(int argc, char** argv) {
    const int Bn = argc > 1 ? atoi(argv[1]) : 51200, dw = argc > 2 ? atoi(argv[2]) : 300, de = argc > 3 ? atoi(argv[3]) : 256;
    float *phrase, *T, *pre, *dx, *gphrase, *ref, *rowsq, *rowsq_ref, *bias; double *cs, *cs_ref;
    CK(hipMalloc(&phrase, (size_t)Bn * dw * 4)); CK(hipMalloc(&T, dw * de * 4)); CK(hipMalloc(&pre, (size_t)Bn * de * 4)); CK(hipMalloc(&dx, (size_t)Bn * de * 4));
    CK(hipMalloc(&gphrase, (size_t)Bn * dw * 4)); CK(hipMalloc(&ref, (size_t)Bn * dw * 4)); CK(hipMalloc(&rowsq, (size_t)Bn * 32 * 4)); CK(hipMalloc(&rowsq_ref, (size_t)Bn * 32 * 4));
    CK(hipMalloc(&cs, 2 * 512 * 8)); CK(hipMalloc(&cs_ref, 2 * 512 * 8)); CK(hipMalloc(&bias, 512 * 4));
    fill_rand<<<2048, 256>>>(phrase, (size_t)Bn * dw, 1, 1.f); fill_rand<<<64, 256>>>(T, dw * de, 2, 0.1f); fill_rand<<<2048, 256>>>(dx, (size_t)Bn * de, 3, 1.f);
    fill_rand<<<1, 256>>>(bias, 512, 4, 1.f);
    CK(hipDeviceSynchronize());
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* nm, double flops, auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipStreamSynchronize(s));
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) fn(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 10); }
        printf("%-44s %.1f us  %.1f TF/s\n", nm, best * 1e3, flops / (best * 1e-3) / 1e12);
    };
    auto cmp = [&](const char* nm, const float* x, const float* y, size_t n) {
        std::vector<float> hx(n), hy(n); CK(hipMemcpy(hx.data(), x, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; size_t bad = 0; for (size_t i = 0; i < n; ++i) { const double d = std::fabs((double)hx[i] - hy[i]); if (!(d <= 1e30)) ++bad; md = std::max(md, d); mx = std::max(mx, (double)std::fabs(hy[i])); }
        printf("   %-30s max|diff| %.3e (max|ref| %.3e) nonfinite %zu\n", nm, md, mx, bad);
    };
    auto cmpd = [&](const char* nm, const double* x, const double* y, size_t n) {
        std::vector<double> hx(n), hy(n); CK(hipMemcpy(hx.data(), x, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy.data(), y, n * 8, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; for (size_t i = 0; i < n; ++i) { md = std::max(md, std::fabs(hx[i] - hy[i])); mx = std::max(mx, std::fabs(hy[i])); }
        printf("   %-30s max|diff| %.3e (max|ref| %.3e)\n", nm, md, mx);
    };
    const double F = 2.0 * Bn * dw * de;
    int parts = 0;
    // ---- forward: pre[B][de] = phrase[B][dw] · T[dw][de] (+ column sums)
    gemm_set_panel_enabled(false);

    CK(hipMemsetAsync(cs_ref, 0, 2 * 512 * 8, s));
    launch_gemm(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, nullptr, 1, 0, s, cs_ref);
    CK(hipMemcpyAsync(ref, pre, (size_t)Bn * de * 4, hipMemcpyDeviceToDevice, s));
    timeit("fwd tiled 128x128 (+colstats)", F, [&] { launch_gemm(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, nullptr, 1, 0, s, cs); });
    CK(hipMemsetAsync(pre, 0, (size_t)Bn * de * 4, s)); CK(hipMemsetAsync(cs, 0, 2 * 512 * 8, s));
    bool ok = launch_gemm_tstat(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, nullptr, s, cs, nullptr, 0.f, &parts);
    printf("fwd tstat launched: %d\n", (int)ok);
    CK(hipStreamSynchronize(s));
    cmp("fwd tstat vs tiled", pre, ref, (size_t)Bn * de);
    cmpd("fwd colstats", cs, cs_ref, 2 * de);
    timeit("fwd tstat (+colstats)", F, [&] { launch_gemm_tstat(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, nullptr, s, cs, nullptr, 0.f, &parts); });
    timeit("fwd tstat (no stats)", F, [&] { launch_gemm_tstat(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, nullptr, s, nullptr, nullptr, 0.f, &parts); });
#ifdef NVSM_TSTAT_DBG
    for (int d : {1, 2, 3}) { g_tstat_dbg = d; char nm[64]; snprintf(nm, 64, "fwd tstat dbg=%d (1 noloadA 2 noepilogue)", d);
        timeit(nm, F, [&] { launch_gemm_tstat(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, nullptr, s, nullptr, nullptr, 0.f, &parts); }); }
    g_tstat_dbg = 0;
#endif
    // with bias
    launch_gemm(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, bias, 1, 0, s);
    CK(hipMemcpyAsync(ref, pre, (size_t)Bn * de * 4, hipMemcpyDeviceToDevice, s));
    launch_gemm_tstat(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, bias, s, nullptr, nullptr, 0.f, &parts);
    cmp("fwd tstat+bias vs tiled", pre, ref, (size_t)Bn * de);
    // ---- bwd_x: gphrase[B][dw] = 0.1 · dx[B][de] · T^T  (T stored [dw][de] = B as [N][K]) + row sums of squares
    const int tparts = gemm_rowsq_parts(dw);
    launch_gemm(0, 1, dx, T, gphrase, Bn, dw, de, de, de, dw, 0.1f, nullptr, 1, 0, s, nullptr, rowsq_ref, 0.5f);
    launch_sum_parts(rowsq_ref, tparts, Bn, rowsq_ref + (size_t)31 * Bn, Bn, s);
    CK(hipMemcpyAsync(ref, gphrase, (size_t)Bn * dw * 4, hipMemcpyDeviceToDevice, s));
    timeit("bwd_x tiled 128x128 (+rowsq)", F, [&] { launch_gemm(0, 1, dx, T, gphrase, Bn, dw, de, de, de, dw, 0.1f, nullptr, 1, 0, s, nullptr, rowsq, 0.5f); });
    CK(hipMemsetAsync(gphrase, 0, (size_t)Bn * dw * 4, s));
    ok = launch_gemm_tstat(0, 1, dx, T, gphrase, Bn, dw, de, de, de, dw, 0.1f, nullptr, s, nullptr, rowsq, 0.5f, &parts);
    printf("bwd_x tstat launched: %d, rowsq parts %d\n", (int)ok, parts);
    if (ok) launch_sum_parts(rowsq, parts, Bn, rowsq + (size_t)31 * Bn, Bn, s);
    CK(hipStreamSynchronize(s));
    cmp("bwd_x tstat vs tiled", gphrase, ref, (size_t)Bn * dw);
    cmp("bwd_x rowsq", rowsq + (size_t)31 * Bn, rowsq_ref + (size_t)31 * Bn, Bn);
    timeit("bwd_x tstat (+rowsq)", F, [&] { launch_gemm_tstat(0, 1, dx, T, gphrase, Bn, dw, de, de, de, dw, 0.1f, nullptr, s, nullptr, rowsq, 0.5f, &parts); });
    timeit("bwd_x tstat (no rowsq)", F, [&] { launch_gemm_tstat(0, 1, dx, T, gphrase, Bn, dw, de, de, de, dw, 0.1f, nullptr, s, nullptr, nullptr, 0.5f, &parts); });
    return 0;
}
