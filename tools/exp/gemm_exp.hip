// Experiment harness (not product code): the three projection GEMM shapes, tiled kernel vs panel kernel variants.
#define NVSM_GEMM_DBG 1
#include "../../cunvsm_amd/csrc/gather_gemm.hip"
#include "../../cunvsm_amd/csrc/gemm_panel.hip"
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
static int g_dbg = 0;
template <int AL, int BL, int TM, int TN, int BK>
void run_panel(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int mpanels, int npanels, int slabs, int ksl, size_t cstride, hipStream_t s) {
    PanelArgs g; g.dbg = g_dbg; g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = 1.f; g.bias_n = nullptr; g.colstats = nullptr;
    g.mpanels = mpanels; g.npanels = npanels; g.slabs = slabs; g.k_split_len = slabs > 1 ? ksl : K; g.c_split_stride = cstride;
    launch_panel<AL, BL, TM, TN, BK>(g, s);
}
int main() {
    const int Bn = 51200, dw = 300, de = 256;
    float *phrase, *T, *pre, *dx, *gphrase, *gT, *part, *ref;
    CK(hipMalloc(&phrase, (size_t)Bn * dw * 4)); CK(hipMalloc(&T, dw * de * 4)); CK(hipMalloc(&pre, (size_t)Bn * de * 4)); CK(hipMalloc(&dx, (size_t)Bn * de * 4));
    CK(hipMalloc(&gphrase, (size_t)Bn * dw * 4)); CK(hipMalloc(&gT, dw * de * 4)); CK(hipMalloc(&part, (size_t)512 * dw * de * 4)); CK(hipMalloc(&ref, (size_t)Bn * dw * 4));
    fill_rand<<<2048, 256>>>(phrase, (size_t)Bn * dw, 1, 1.f); fill_rand<<<64, 256>>>(T, dw * de, 2, 0.1f); fill_rand<<<2048, 256>>>(dx, (size_t)Bn * de, 3, 1.f);
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
        double md = 0, mx = 0; for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)std::fabs(hx[i] - hy[i])); mx = std::max(mx, (double)std::fabs(hx[i])); }
        printf("   %-30s max|diff| %.3e (max|x| %.3e)\n", nm, md, mx);
    };
    const double F = 2.0 * Bn * dw * de;
    // ---- forward: pre[B][de] = phrase[B][dw] · T[dw][de]
    gemm_set_panel_enabled(false);
    timeit("fwd tiled 128x128", F, [&] { launch_gemm(0, 0, phrase, T, pre, Bn, de, dw, dw, de, de, 1.f, nullptr, 1, 0, s); });
    CK(hipMemcpyAsync(ref, pre, (size_t)Bn * de * 4, hipMemcpyDeviceToDevice, s));
    timeit("fwd panel TM13 TN4 BK32", F, [&] { run_panel<0, 0, 13, 4, 32>(phrase, T, pre, Bn, de, dw, dw, de, de, (Bn + 207) / 208, 1, 1, 0, 0, s); });
    cmp("fwd panel vs tiled", pre, ref, (size_t)Bn * de);
    for (int d : {1, 2, 4, 3, 5, 7}) { g_dbg = d; char nm[64]; snprintf(nm, 64, "fwd panel TM13 dbg=%d (1 noload 2 nostore 4 nolds)", d);
        timeit(nm, F, [&] { run_panel<0, 0, 13, 4, 32>(phrase, T, pre, Bn, de, dw, dw, de, de, (Bn + 207) / 208, 1, 1, 0, 0, s); }); }
    g_dbg = 0;
    timeit("fwd panel TM13 TN4 BK16", F, [&] { run_panel<0, 0, 13, 4, 16>(phrase, T, pre, Bn, de, dw, dw, de, de, (Bn + 207) / 208, 1, 1, 0, 0, s); });
    timeit("fwd panel TM7 TN4 BK32 (458 blk)", F, [&] { run_panel<0, 0, 7, 4, 32>(phrase, T, pre, Bn, de, dw, dw, de, de, (Bn + 111) / 112, 1, 1, 0, 0, s); });
    cmp("fwd panel7 vs tiled", pre, ref, (size_t)Bn * de);
    // ---- bwd_x: gphrase[B][dw] = dx[B][de] · T^T  (T stored [dw][de] = B as [N][K])
    timeit("bwd_x tiled 128x128", F, [&] { launch_gemm(0, 1, dx, T, gphrase, Bn, dw, de, de, de, dw, 1.f, nullptr, 1, 0, s); });
    CK(hipMemcpyAsync(ref, gphrase, (size_t)Bn * dw * 4, hipMemcpyDeviceToDevice, s));
    timeit("bwd_x panel TM13 TN5 BK16 (spills)", F, [&] { run_panel<0, 1, 13, 5, 16>(dx, T, gphrase, Bn, dw, de, de, de, dw, (Bn + 207) / 208, 1, 1, 0, 0, s); });
    cmp("bwd_x panel13 vs tiled", gphrase, ref, (size_t)Bn * dw);
    timeit("bwd_x panel TM7 TN5 BK32 (458 blk)", F, [&] { run_panel<0, 1, 7, 5, 32>(dx, T, gphrase, Bn, dw, de, de, de, dw, (Bn + 111) / 112, 1, 1, 0, 0, s); });
    cmp("bwd_x panel7 vs tiled", gphrase, ref, (size_t)Bn * dw);
    timeit("bwd_x panel TM12 TN5 BK16 (267 blk)", F, [&] { run_panel<0, 1, 12, 5, 16>(dx, T, gphrase, Bn, dw, de, de, de, dw, (Bn + 191) / 192, 1, 1, 0, 0, s); });
    // ---- bwd_T: gT[dw][de] = phrase^T · dx, split-K
    const size_t stride = (size_t)dw * de;
    timeit("bwd_T tiled split128 + reduce", F, [&] { launch_gemm(1, 0, phrase, dx, part, dw, de, Bn, dw, de, de, 1.f, nullptr, 128, stride, s);
        launch_splitk_reduce(part, gemm_split_k_slabs(Bn, 128), stride, gT, stride, s); });
    CK(hipMemcpyAsync(ref, gT, stride * 4, hipMemcpyDeviceToDevice, s));
    timeit("bwd_T tiled split128 (gemm only)", F, [&] { launch_gemm(1, 0, phrase, dx, part, dw, de, Bn, dw, de, de, 1.f, nullptr, 128, stride, s); });
    timeit("reduce 128 slabs only", 0, [&] { launch_splitk_reduce(part, 128, stride, gT, stride, s); });
    timeit("bwd_T panel TM10 TN4 x128 slabs (256 blk)", F, [&] { run_panel<1, 0, 10, 4, 32>(phrase, dx, part, dw, de, Bn, dw, de, de, 2, 1, 128, 400, stride, s); });
    launch_splitk_reduce(part, 128, stride, gT, stride, s);
    cmp("bwd_T panel vs tiled", gT, ref, stride);
    timeit("bwd_T panel TM19 TN4 x256 slabs (256 blk)", F, [&] { run_panel<1, 0, 19, 4, 32>(phrase, dx, part, dw, de, Bn, dw, de, de, 1, 1, 256, 200, stride, s); });
    launch_splitk_reduce(part, 256, stride, gT, stride, s);
    cmp("bwd_T panel19 vs tiled", gT, ref, stride);
    timeit("bwd_T panel TM10 TN4 x256 slabs (512 blk)", F, [&] { run_panel<1, 0, 10, 4, 32>(phrase, dx, part, dw, de, Bn, dw, de, de, 2, 1, 256, 200, stride, s); });
    return 0;
}
