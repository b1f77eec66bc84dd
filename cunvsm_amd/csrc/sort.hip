// Batch → row order: stable LSD radix sort of (table row, entry id) pairs via rocPRIM/hipCUB.
// Plumbing, not arithmetic: it only permutes entry ids so that update.hip's row passes can own rows.
#include "kernels.h"

#include <hipcub/hipcub.hpp>
#include <stdexcept>

namespace cunvsm {

size_t sort_pairs_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, static_cast<const int*>(nullptr), static_cast<int*>(nullptr),
                                       static_cast<const int*>(nullptr), static_cast<int*>(nullptr),
                                       static_cast<int>(n), 0, bits, nullptr);
    return bytes;
}

void sort_pairs(void* temp, size_t temp_bytes, const int* keys_in, int* keys_out, const int* vals_in, int* vals_out,
                int64_t n, int bits, hipStream_t s) {
    if (n <= 0) return;
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                                            static_cast<int>(n), 0, bits, s);
    if (e != hipSuccess) throw std::runtime_error(std::string("radix sort failed: ") + hipGetErrorString(e));
}

}  // namespace cunvsm
