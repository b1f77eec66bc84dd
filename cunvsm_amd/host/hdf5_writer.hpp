// write_to_hdf5 (cpp/hdf5.cu:26-53, include/cuNVSM/lse_hdf5_inl.h:4-28): one float32 little-endian dataset per
// parameter, dims {cols, rows} of the reference's column-major device_matrix — i.e. [num_objects][dim] for the
// embedding tables, [word_dim][entity_dim] for the projection and [1][entity_dim] for the bias, exactly what
// py/nvsm/base.py:182-240 asserts — in a file created with H5F_ACC_EXCL (fails if it already exists).
// libhdf5 (>= 1.10) is resolved at run time with dlopen, so the trainer itself links against nothing but libc,
// libstdc++ and libcunvsm_amd.so.
#pragma once

#include <string>
#include <vector>

namespace nvsm_host {

struct Hdf5Dataset {
    std::string name;
    unsigned long long dim0, dim1;   // as written: {cols, rows} of the reference matrix
    const float* data;               // dim0 * dim1 floats, row-major [dim0][dim1]
};

// Throws FatalError (file exists, library missing, write error).
void write_hdf5(const std::string& filename, const std::vector<Hdf5Dataset>& datasets);
// Loads libhdf5 and initialises it now (the reference links it, so its loader pays this before main(); here it would otherwise
// land inside the first model dump — on a cold box 0.9 s in the middle of the second epoch's batches-per-second figure).
void hdf5_preload();

}  // namespace nvsm_host
