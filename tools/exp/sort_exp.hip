// Experiment harness (not product code): batch -> row-order sort variants.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

using force_onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
template <int BITS, int IPT, int HIPT = 8>
using os_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, HIPT>, rocprim::kernel_config<1024, IPT>, BITS,
                                        rocprim::block_radix_rank_algorithm::match>, 0>;

int main() {
    struct Case { const char* name; int n; int rows; bool zipf; };
    Case cases[] = {{"entities uniform", 870400, 100000, false}, {"words zipf", 512000, 50000, true}, {"words uniform", 512000, 50000, false},
                    {"entities 2M rows", 870400, 2000000, false}, {"ents B=4096", 69632, 100000, false}, {"words B=4096 200k", 40960, 200000, true}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& c : cases) {
        std::vector<int> hk(c.n), hv(c.n);
        std::mt19937 rng(1);
        if (c.zipf) {
            std::vector<double> cdf(c.rows); double acc = 0; for (int i = 0; i < c.rows; ++i) { acc += 1.0 / (i + 1); cdf[i] = acc; }
            std::uniform_real_distribution<double> U(0, acc);
            for (int i = 0; i < c.n; ++i) hk[i] = int(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin());
        } else { std::uniform_int_distribution<int> U(0, c.rows - 1); for (int i = 0; i < c.n; ++i) hk[i] = U(rng); }
        for (int i = 0; i < c.n; ++i) hv[i] = i;
        int bits = 1; while ((1ll << bits) < c.rows) ++bits;
        int *k_in, *k_out, *v_in, *v_out; void* tmp; size_t tb1 = 0, tb2 = 0;
        CK(hipMalloc(&k_in, c.n * 4)); CK(hipMalloc(&k_out, c.n * 4)); CK(hipMalloc(&v_in, c.n * 4)); CK(hipMalloc(&v_out, c.n * 4));
        CK(hipMemcpy(k_in, hk.data(), c.n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v_in, hv.data(), c.n * 4, hipMemcpyHostToDevice));
        CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, k_in, k_out, v_in, v_out, c.n, 0, bits, s));
        CK((rocprim::radix_sort_pairs<force_onesweep>(nullptr, tb2, k_in, k_out, v_in, v_out, (size_t)c.n, 0, bits, s)));
        CK(hipMalloc(&tmp, std::max(tb1, tb2) + 256));
        auto timeit = [&](const char* nm, auto fn) {
            for (int i = 0; i < 3; ++i) fn();
            CK(hipStreamSynchronize(s));
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) fn(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 10);
            }
            printf("%-20s %-28s n=%d bits=%d: %.1f us\n", c.name, nm, c.n, bits, best * 1e3);
        };
        timeit("hipcub default", [&] { CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb1, k_in, k_out, v_in, v_out, c.n, 0, bits, s)); });
        std::vector<int> r1(c.n), r2(c.n);
        CK(hipStreamSynchronize(s)); CK(hipMemcpy(r1.data(), v_out, c.n * 4, hipMemcpyDeviceToHost));
        timeit("rocprim onesweep forced", [&] { CK((rocprim::radix_sort_pairs<force_onesweep>(tmp, tb2, k_in, k_out, v_in, v_out, (size_t)c.n, 0, bits, s))); });
        CK(hipStreamSynchronize(s)); CK(hipMemcpy(r2.data(), v_out, c.n * 4, hipMemcpyDeviceToHost));
        printf("   same permutation: %s (tmp %zu / %zu bytes)\n", r1 == r2 ? "yes" : "NO", tb1, tb2);
        auto variant = [&](const char* nm, auto cfg_tag) {
            using CFG = decltype(cfg_tag);
            size_t tb = 0;
            CK((rocprim::radix_sort_pairs<CFG>(nullptr, tb, k_in, k_out, v_in, v_out, (size_t)c.n, 0, bits, s)));
            void* t2; CK(hipMalloc(&t2, tb + 256));
            timeit(nm, [&] { CK((rocprim::radix_sort_pairs<CFG>(t2, tb, k_in, k_out, v_in, v_out, (size_t)c.n, 0, bits, s))); });
            std::vector<int> r3(c.n); CK(hipStreamSynchronize(s)); CK(hipMemcpy(r3.data(), v_out, c.n * 4, hipMemcpyDeviceToHost));
            if (r3 != r1) printf("   %s: DIFFERENT permutation\n", nm);
            hipFree(t2);
        };
        variant("onesweep 8b ipt8", os_cfg<8, 8>{});
        variant("onesweep 9b ipt8", os_cfg<9, 8>{});
        variant("onesweep 9b ipt4", os_cfg<9, 4>{});
        variant("onesweep 10b ipt4", os_cfg<10, 4>{});
        variant("onesweep 10b ipt8", os_cfg<10, 8>{});
        variant("onesweep 6b ipt8", os_cfg<6, 8>{});
        hipFree(k_in); hipFree(k_out); hipFree(v_in); hipFree(v_out); hipFree(tmp);
    }
    return 0;
}
