// extern "C" surface of libcunvsm_amd_testhooks.so — include/cunvsm_amd_test_hooks.h: entry points for the UNIT TESTS of single
// kernels (GEMM variants, the row sort, the gather-mean), two profiling aids (a spin kernel in front of a step, the three-launch form
// of the table passes) and the timing loops of tools/exp/. Not part of the drop-in surface and not in libcunvsm_amd.so:
// this library is loaded BEHIND the product library (RTLD_GLOBAL) and calls the product's own launchers — it carries no kernel and
// no copy of the engine, so what the hooks exercise is the code the product ships.
#include <algorithm>
#include <cstring>
#include <vector>
#include <string>

#include "c_api_internal.h"
#include "../../include/cunvsm_amd_test_hooks.h"

using cunvsm::Error;
using cunvsm::guarded_hook;
using cunvsm::guarded_on;

extern "C" {

int nvsm_debug_set_table_pass_form(int one_launch) { cunvsm::set_table_pass_one_launch(one_launch != 0); return NVSM_OK; }
int nvsm_debug_delay(nvsm_model* m, int microseconds) { NVSM_REQUIRE(m); return guarded_on(m, [&] { m->impl.debug_delay(microseconds); }); }

// ---- debug hooks (tests only) ----
int nvsm_debug_gemm(int variant, int M, int N, int K, const float* hostA, const float* hostB, float* hostC) {
    NVSM_REQUIRE(hostA); NVSM_REQUIRE(hostB); NVSM_REQUIRE(hostC);
    return guarded_hook([&] {
        const int al = (variant >> 1) & 1, bl = variant & 1;
        const bool exact = (variant >> 30) & 1;               // bit 30: the exact-fp32 tiled kernel even where the planes kernel covers the shape
        const bool wave = (variant >> 29) & 1;                // bit 29: the projection-gradient shape on the wave-sized kernel (gemm_dtw.hip)
        const int split = (variant & 0x1fffffff) >> 2;        // variant bits: [exact << 30 | wave << 29 | split_k want << 2 | a_layout << 1 | b_layout]
        cunvsm::DevBuf<float> A, B, C, P;
        A.alloc(static_cast<size_t>(M) * K); B.alloc(static_cast<size_t>(K) * N); C.alloc(static_cast<size_t>(M) * N);
        NVSM_HIP_CHECK(hipMemcpy(A.p, hostA, A.n * sizeof(float), hipMemcpyHostToDevice));
        NVSM_HIP_CHECK(hipMemcpy(B.p, hostB, B.n * sizeof(float), hipMemcpyHostToDevice));
        const int lda = al ? M : K, ldb = bl ? K : N;
        if (split >= 1 && al == 1 && bl == 0 && wave) {
            if (!cunvsm::gemm_dtw_covers(M, N, K)) throw Error(NVSM_ERR_UNSUPPORTED, "gemm_dtw does not cover the shape");
            const int slabs = cunvsm::gemm_dtw_slabs(K, split);
            P.alloc(static_cast<size_t>(slabs) * M * N);
            if (!cunvsm::launch_gemm_dtw(A.p, B.p, slabs == 1 ? C.p : P.p, M, N, K, lda, ldb, split, nullptr)) throw Error(NVSM_ERR_UNSUPPORTED, "gemm_dtw refused a covered shape");
            if (slabs > 1) cunvsm::launch_splitk_reduce(P.p, slabs, static_cast<size_t>(M) * N, C.p, static_cast<int64_t>(M) * N, nullptr);
            NVSM_HIP_CHECK(hipDeviceSynchronize());
        } else if (split >= 1 && al == 1 && bl == 0 && !exact && cunvsm::gemm_dt_covers(M, N, K)) {
            // the projection-gradient shape: the split-bf16 split-K kernel the model uses for it (gemm_dt.hip)
            const int slabs = cunvsm::gemm_dt_slabs(K, split);
            P.alloc(static_cast<size_t>(slabs) * M * N);
            if (!cunvsm::launch_gemm_dt(A.p, B.p, P.p, M, N, K, lda, ldb, split, nullptr)) throw Error(NVSM_ERR_UNSUPPORTED, "gemm_dt refused a covered shape");
            cunvsm::launch_splitk_reduce(P.p, slabs, static_cast<size_t>(M) * N, C.p, static_cast<int64_t>(M) * N, nullptr);
            NVSM_HIP_CHECK(hipDeviceSynchronize());
        } else if (split > 1) {
            const int slabs = cunvsm::gemm_split_k_slabs(K, split);
            P.alloc(static_cast<size_t>(slabs) * M * N);
            cunvsm::launch_gemm(al, bl, A.p, B.p, P.p, M, N, K, lda, ldb, N, 1.f, nullptr, split, static_cast<size_t>(M) * N, nullptr);
            cunvsm::launch_splitk_reduce(P.p, slabs, static_cast<size_t>(M) * N, C.p, static_cast<int64_t>(M) * N, nullptr);
        } else {
            cunvsm::DevBuf<char> planes, rplanes;
            planes.alloc(cunvsm::gemm_split_planes_bytes(N, K));
            rplanes.alloc(cunvsm::gemm_rsplit_planes_bytes(N, K));
            cunvsm::GemmSplitWs sws{planes.p, planes.n, false, rplanes.p, rplanes.n, false};
            cunvsm::launch_gemm(al, bl, A.p, B.p, C.p, M, N, K, lda, ldb, N, 1.f, nullptr, 1, 0, nullptr, nullptr, nullptr, 0.f, nullptr, false,
                                nullptr, &sws);
            NVSM_HIP_CHECK(hipDeviceSynchronize());
        }
        NVSM_HIP_CHECK(hipDeviceSynchronize());
        NVSM_HIP_CHECK(hipMemcpy(hostC, C.p, C.n * sizeof(float), hipMemcpyDeviceToHost));
    });
}

// average milliseconds of one launch_gemm of the batch-sized products on device-resident operands (A [M][K], B per b_layout):
// extras bit 0 = ordered column statistics (the forward product), bit 1 = row sums of squares (the backward one)
int nvsm_debug_gemm_time(int b_layout, int M, int N, int K, int extras, int repeats, float* avg_ms) {
    NVSM_REQUIRE(avg_ms);
    return guarded_hook([&] {
        cunvsm::DevBuf<float> A, B, C, rowsq, part;
        cunvsm::DevBuf<double> stats, part2;
        cunvsm::DevBuf<int> arrive;
        A.alloc(static_cast<size_t>(M) * K); B.alloc(static_cast<size_t>(K) * N); C.alloc(static_cast<size_t>(M) * N);
        {   // operands with all 24 significant bits in use (zeros would flatter a kernel: less switching, higher clocks)
            std::vector<float> h(std::max(A.n, B.n));
            uint32_t x = 12345u;
            auto fill = [&](cunvsm::DevBuf<float>& d, float scale) {
                for (size_t i = 0; i < d.n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (static_cast<float>(x >> 8) * (1.f / 8388608.f) - 1.f) * scale; }
                NVSM_HIP_CHECK(hipMemcpy(d.p, h.data(), d.n * sizeof(float), hipMemcpyHostToDevice));
            };
            fill(A, 0.5f); fill(B, 0.1f);
        }
        cunvsm::GridSumWs ws{};
        if (extras & 1) {
            ws.colgroups = 8; ws.contrib_cap = M / 32 + 512; ws.width_cap = 2 * (N > 160 ? N : 160); ws.groups_cap = ws.contrib_cap / 16 + 1; ws.fan = 16;
            part.alloc(static_cast<size_t>(ws.colgroups) * ws.contrib_cap * ws.width_cap);
            part2.alloc(static_cast<size_t>(ws.colgroups) * ws.groups_cap * ws.width_cap);
            arrive.alloc(static_cast<size_t>(ws.colgroups) * (ws.groups_cap + 1), true);
            stats.alloc(2 * static_cast<size_t>(N), true);
            ws.part = part.p; ws.part2 = part2.p; ws.arrive = arrive.p;
        }
        if (extras & 2) rowsq.alloc(static_cast<size_t>(M) * cunvsm::gemm_rowsq_parts(N));
        hipStream_t s;
        NVSM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipEvent_t e0, e1;
        NVSM_HIP_CHECK(hipEventCreate(&e0)); NVSM_HIP_CHECK(hipEventCreate(&e1));
        const int ldb = b_layout ? K : N;
        int parts = 0;
        cunvsm::DevBuf<char> planes, rplanes;
        planes.alloc(cunvsm::gemm_split_planes_bytes(N, K));
        rplanes.alloc(cunvsm::gemm_rsplit_planes_bytes(N, K));
        cunvsm::GemmSplitWs sws{planes.p, planes.n, false, rplanes.p, rplanes.n, false};
        auto go = [&] {
            sws.ready = (extras & 4) != 0 && sws.ready;          // extras bit 2: the planes of B stay valid between launches
            sws.rready = (extras & 4) != 0 && sws.rready;
            cunvsm::launch_gemm(0, b_layout, A.p, B.p, C.p, M, N, K, K, ldb, N, 1.f, nullptr, 1, 0, s, (extras & 1) ? stats.p : nullptr,
                                (extras & 2) ? rowsq.p : nullptr, 1.f, &parts, false, (extras & 1) ? &ws : nullptr, &sws);
        };
        for (int i = 0; i < 3; ++i) go();
        NVSM_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < repeats; ++i) go();
        NVSM_HIP_CHECK(hipEventRecord(e1, s));
        NVSM_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / static_cast<float>(repeats > 0 ? repeats : 1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    });
}

// the dT product alone on device-resident operands (phrase [rows][M], dx [rows][N]): average ms of the split-K kernel and of
// the reduce behind it. which 0 = gemm_dt (split-bf16), 2 = the tiled exact-fp32 kernel
int nvsm_debug_dt_time(int M, int N, int rows, int slabs, int repeats, int which, float* kernel_ms, float* reduce_ms) {
    NVSM_REQUIRE(kernel_ms); NVSM_REQUIRE(reduce_ms);
    return guarded_hook([&] {
        cunvsm::DevBuf<float> A, B, C, P;
        A.alloc(static_cast<size_t>(rows) * M); B.alloc(static_cast<size_t>(rows) * N); C.alloc(static_cast<size_t>(M) * N);
        {
            std::vector<float> h(std::max(A.n, B.n));
            uint32_t x = 4242u;
            auto fill = [&](cunvsm::DevBuf<float>& d, float scale) {
                for (size_t i = 0; i < d.n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (static_cast<float>(x >> 8) * (1.f / 8388608.f) - 1.f) * scale; }
                NVSM_HIP_CHECK(hipMemcpy(d.p, h.data(), d.n * sizeof(float), hipMemcpyHostToDevice));
            };
            fill(A, 0.5f); fill(B, 0.1f);
        }
        hipStream_t s;
        NVSM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if (which != 0 && which != 1 && which != 2) throw Error(NVSM_ERR_INVALID_ARGUMENT, "which: 0 = split-bf16 kernel, 1 = its wave-sized form, 2 = tiled fp32 kernel");
        const int ns = which == 0 ? cunvsm::gemm_dt_slabs(rows, slabs) : (which == 1 ? cunvsm::gemm_dtw_slabs(rows, slabs) : cunvsm::gemm_split_k_slabs(rows, slabs));
        const size_t stride = static_cast<size_t>(M) * N;
        P.alloc(static_cast<size_t>(std::max(ns, 1)) * stride);
        auto product = [&] {
            bool ok = true;
            if (which == 0) ok = cunvsm::launch_gemm_dt(A.p, B.p, P.p, M, N, rows, M, N, slabs, s);
            else if (which == 1) ok = cunvsm::launch_gemm_dtw(A.p, B.p, P.p, M, N, rows, M, N, slabs, s);
            else cunvsm::launch_gemm(1, 0, A.p, B.p, P.p, M, N, rows, M, N, N, 1.f, nullptr, slabs, stride, s);
            if (!ok) throw Error(NVSM_ERR_UNSUPPORTED, "the dT kernel refused the shape");
        };
        auto reduce = [&] { cunvsm::launch_splitk_reduce(P.p, ns, stride, C.p, static_cast<int64_t>(stride), s); };
        hipEvent_t e0, e1, e2;
        NVSM_HIP_CHECK(hipEventCreate(&e0)); NVSM_HIP_CHECK(hipEventCreate(&e1)); NVSM_HIP_CHECK(hipEventCreate(&e2));
        for (int i = 0; i < 3; ++i) { product(); reduce(); }
        const int reps = repeats > 0 ? repeats : 1;
        NVSM_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) product();
        NVSM_HIP_CHECK(hipEventRecord(e1, s));
        for (int i = 0; i < reps; ++i) reduce();
        NVSM_HIP_CHECK(hipEventRecord(e2, s));
        NVSM_HIP_CHECK(hipEventSynchronize(e2));
        float a = 0.f, b = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&a, e0, e1)); NVSM_HIP_CHECK(hipEventElapsedTime(&b, e1, e2));
        *kernel_ms = a / reps; *reduce_ms = b / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipStreamDestroy(s);
    });
}

int nvsm_debug_sort(int64_t n, int bits, const int32_t* keys, int32_t* keys_out, int32_t* vals_out, int repeats, float* avg_ms) {
    NVSM_REQUIRE(keys); NVSM_REQUIRE(keys_out); NVSM_REQUIRE(vals_out);
    return guarded_hook([&] {
        if (n <= 0) return;
        cunvsm::DevBuf<int> K, KO, VO;
        cunvsm::DevBuf<char> tmp;
        K.alloc(n); KO.alloc(n); VO.alloc(n);
        const size_t tb = cunvsm::sort_pairs_temp_bytes(n, bits);
        tmp.alloc(tb, true);
        int* err = nullptr;
        NVSM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&err), sizeof(int), hipHostMallocDefault));
        *err = 0;
        uint64_t epoch = 0;
        NVSM_HIP_CHECK(hipMemcpy(K.p, keys, n * sizeof(int), hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        NVSM_HIP_CHECK(hipEventCreate(&e0)); NVSM_HIP_CHECK(hipEventCreate(&e1));
        cunvsm::sort_pairs(tmp.p, tb, &epoch, K.p, KO.p, nullptr, VO.p, n, bits, err, nullptr);      // warm-up
        NVSM_HIP_CHECK(hipEventRecord(e0, nullptr));
        const int reps = repeats > 0 ? repeats : 1;
        for (int r = 0; r < reps; ++r)      // repeats: the arrival counter keeps growing across calls
            cunvsm::sort_pairs(tmp.p, tb, &epoch, K.p, KO.p, nullptr, VO.p, n, bits, err, nullptr);
        NVSM_HIP_CHECK(hipEventRecord(e1, nullptr));
        NVSM_HIP_CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        NVSM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (avg_ms) *avg_ms = ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        const int code = *err;
        (void)hipHostFree(err);
        if (code) throw Error(NVSM_ERR_DEVICE, "sort reported an error");
        NVSM_HIP_CHECK(hipMemcpy(keys_out, KO.p, n * sizeof(int), hipMemcpyDeviceToHost));
        NVSM_HIP_CHECK(hipMemcpy(vals_out, VO.p, n * sizeof(int), hipMemcpyDeviceToHost));
    });
}

int nvsm_debug_gather_mean(int64_t num_rows, int dim, const float* table, const int64_t* idx, const float* wts,
                           int window, int64_t num_out, float* out) {
    NVSM_REQUIRE(table); NVSM_REQUIRE(idx); NVSM_REQUIRE(out);
    return guarded_hook([&] {
        cunvsm::DevBuf<float> T, W, O;
        cunvsm::DevBuf<int64_t> I64;
        cunvsm::DevBuf<int> I;
        T.alloc(num_rows * dim); O.alloc(num_out * dim); I64.alloc(num_out * window); I.alloc(num_out * window);
        NVSM_HIP_CHECK(hipMemcpy(T.p, table, T.n * sizeof(float), hipMemcpyHostToDevice));
        NVSM_HIP_CHECK(hipMemcpy(I64.p, idx, I64.n * sizeof(int64_t), hipMemcpyHostToDevice));
        if (wts) { W.alloc(num_out * window); NVSM_HIP_CHECK(hipMemcpy(W.p, wts, W.n * sizeof(float), hipMemcpyHostToDevice)); }
        cunvsm::launch_narrow_i64(I64.p, I.p, num_out * window, num_rows, nullptr, 0, nullptr);
        cunvsm::launch_gather_mean(T.p, dim, I.p, wts ? W.p : nullptr, window, num_out, O.p, nullptr);
        NVSM_HIP_CHECK(hipDeviceSynchronize());
        NVSM_HIP_CHECK(hipMemcpy(out, O.p, O.n * sizeof(float), hipMemcpyDeviceToHost));
    });
}


}  // extern "C"
