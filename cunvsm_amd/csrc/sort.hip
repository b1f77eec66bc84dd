// Batch → row order: stable LSD radix sort of (table row, entry id) pairs via rocPRIM.
// Plumbing, not arithmetic: it only permutes entry ids so that update.hip's row passes can own rows.
// rocPRIM's default dispatch picks a merge sort below 1M items (19 merge launches per sort at our sizes: 153 us for
// 870 k pairs); the Onesweep radix path is forced instead. Two configurations (tools/exp/sort_exp.hip; same
// permutation in every case):
//  * small batches (< 256 k pairs): 9-bit digits — row keys have 16-18 bits, i.e. two passes instead of three — and
//    1024 x 4 items per block: 70 k pairs / 17 bits 33 us instead of 89 us. At batch 4096 the words update waits for
//    this chain, so its latency is step time.
//  * large batches: rocPRIM's default Onesweep configuration (870 k pairs / 17 bits 102 us). The 9-bit configuration
//    sorts them in 51 us alone, but the sorts run on the side streams next to the projection GEMM and the loss kernel,
//    and its fatter blocks take more from those than the shorter chain gives back (interleaved A/B of the whole step:
//    1.089 vs 1.080 ms).
#include "kernels.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdexcept>
#include <string>

namespace cunvsm {

using sort_config_large = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
using sort_config_small = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 4>, 9,
                                        rocprim::block_radix_rank_algorithm::match>,
    0>;
constexpr int64_t kSmallSort = 1 << 18;

template <class Config>
static hipError_t sort_call(void* temp, size_t& temp_bytes, const int* keys_in, int* keys_out, const int* vals_in, int* vals_out,
                            int64_t n, int bits, hipStream_t s) {
    return rocprim::radix_sort_pairs<Config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, static_cast<size_t>(n), 0u,
                                             static_cast<unsigned>(bits), s);
}

// workspace for any n' <= n (the batch may be ragged): the larger of the two configurations' needs
size_t sort_pairs_temp_bytes(int64_t n, int bits) {
    size_t a = 0, b = 0;
    (void)sort_call<sort_config_large>(nullptr, a, nullptr, nullptr, nullptr, nullptr, n, bits, nullptr);
    (void)sort_call<sort_config_small>(nullptr, b, nullptr, nullptr, nullptr, nullptr, n < kSmallSort ? n : kSmallSort, bits, nullptr);
    return a > b ? a : b;
}

void sort_pairs(void* temp, size_t temp_bytes, const int* keys_in, int* keys_out, const int* vals_in, int* vals_out,
                int64_t n, int bits, hipStream_t s) {
    if (n <= 0) return;
    const hipError_t e = (n < kSmallSort)
        ? sort_call<sort_config_small>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, bits, s)
        : sort_call<sort_config_large>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, bits, s);
    if (e != hipSuccess) throw std::runtime_error(std::string("radix sort failed: ") + hipGetErrorString(e));
}

}  // namespace cunvsm
