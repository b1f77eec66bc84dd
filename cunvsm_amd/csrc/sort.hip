// Batch → row order: stable LSD radix sort of (table row, entry id) pairs via rocPRIM.
// Plumbing, not arithmetic: it only permutes entry ids so that update.hip's row passes can own rows.
// rocPRIM's default dispatch picks a merge sort below 1M items (19 merge launches per sort at our sizes:
// 155 us for 870 k pairs); the Onesweep radix path is forced instead (103 us / 68 us for the two tables).
#include "kernels.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdexcept>
#include <string>

namespace cunvsm {

using sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

size_t sort_pairs_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<sort_config>(nullptr, bytes, static_cast<const int*>(nullptr), static_cast<int*>(nullptr),
                                                 static_cast<const int*>(nullptr), static_cast<int*>(nullptr),
                                                 static_cast<size_t>(n), 0u, static_cast<unsigned>(bits), nullptr);
    return bytes;
}

void sort_pairs(void* temp, size_t temp_bytes, const int* keys_in, int* keys_out, const int* vals_in, int* vals_out,
                int64_t n, int bits, hipStream_t s) {
    if (n <= 0) return;
    const hipError_t e = rocprim::radix_sort_pairs<sort_config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                                                static_cast<size_t>(n), 0u, static_cast<unsigned>(bits), s);
    if (e != hipSuccess) throw std::runtime_error(std::string("radix sort failed: ") + hipGetErrorString(e));
}

}  // namespace cunvsm
