#include "hdf5_writer.hpp"

#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>

#include "base.hpp"

namespace nvsm_host {
namespace {

// The handful of HDF5 1.10+ C entry points the writer needs (hid_t is int64_t since 1.10), declared here instead
// of pulling a foreign include directory into the build.
typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;
constexpr unsigned kH5F_ACC_EXCL = 0x0004u;
constexpr hid_t kH5P_DEFAULT = 0, kH5S_ALL = 0;
constexpr int kH5T_ORDER_LE = 0;

struct Hdf5Api {
    void* handle = nullptr;
    herr_t (*open)() = nullptr;
    herr_t (*get_libversion)(unsigned*, unsigned*, unsigned*) = nullptr;
    hid_t (*Fcreate)(const char*, unsigned, hid_t, hid_t) = nullptr;
    herr_t (*Fclose)(hid_t) = nullptr;
    hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
    herr_t (*Sclose)(hid_t) = nullptr;
    hid_t (*Tcopy)(hid_t) = nullptr;
    herr_t (*Tset_order)(hid_t, int) = nullptr;
    herr_t (*Tclose)(hid_t) = nullptr;
    hid_t (*Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*) = nullptr;
    herr_t (*Dclose)(hid_t) = nullptr;
    herr_t (*Eset_auto2)(hid_t, void*, void*) = nullptr;
    hid_t* native_float = nullptr;

    static Hdf5Api& get() {
        static Hdf5Api api;
        if (api.handle) return api;
        std::vector<std::string> names;
        if (const char* env = std::getenv("NVSM_HDF5_LIB")) names.push_back(env);
        for (const char* n : {"libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5_serial.so.103", "/opt/conda/lib/libhdf5.so",
                              "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so"})
            names.push_back(n);
        std::string tried;
        for (const std::string& n : names) {
            api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            tried += n + " ";
        }
        if (!api.handle) NVSM_LOG(FATAL) << "cannot load libhdf5 (set NVSM_HDF5_LIB); tried: " << tried;
#define NVSM_H5_SYM(field, sym)                                                             \
        api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym));          \
        if (!api.field) NVSM_LOG(FATAL) << "libhdf5 lacks " << sym;
        NVSM_H5_SYM(open, "H5open") NVSM_H5_SYM(get_libversion, "H5get_libversion") NVSM_H5_SYM(Fcreate, "H5Fcreate")
        NVSM_H5_SYM(Fclose, "H5Fclose") NVSM_H5_SYM(Screate_simple, "H5Screate_simple") NVSM_H5_SYM(Sclose, "H5Sclose")
        NVSM_H5_SYM(Tcopy, "H5Tcopy") NVSM_H5_SYM(Tset_order, "H5Tset_order") NVSM_H5_SYM(Tclose, "H5Tclose")
        NVSM_H5_SYM(Dcreate2, "H5Dcreate2") NVSM_H5_SYM(Dwrite, "H5Dwrite") NVSM_H5_SYM(Dclose, "H5Dclose")
        NVSM_H5_SYM(Eset_auto2, "H5Eset_auto2") NVSM_H5_SYM(native_float, "H5T_NATIVE_FLOAT_g")
#undef NVSM_H5_SYM
        unsigned maj = 0, min = 0, rel = 0;
        api.get_libversion(&maj, &min, &rel);
        if (maj != 1 || min < 10) NVSM_LOG(FATAL) << "libhdf5 " << maj << "." << min << "." << rel << " found; 1.10 or newer is required";
        if (api.open() < 0) NVSM_LOG(FATAL) << "H5open failed";
        return api;
    }
};

}  // namespace

void hdf5_preload() { (void)Hdf5Api::get(); }

void write_hdf5(const std::string& filename, const std::vector<Hdf5Dataset>& datasets) {
    Hdf5Api& h5 = Hdf5Api::get();
    h5.Eset_auto2(0 /* H5E_DEFAULT */, nullptr, nullptr);      // errors are reported through return codes below
    const hid_t file = h5.Fcreate(filename.c_str(), kH5F_ACC_EXCL, kH5P_DEFAULT, kH5P_DEFAULT);
    if (file < 0) NVSM_LOG(FATAL) << "unable to create " << filename << " (H5F_ACC_EXCL: the file must not exist yet)";
    for (const Hdf5Dataset& d : datasets) {
        const hsize_t dims[2] = {d.dim0, d.dim1};
        const hid_t space = h5.Screate_simple(2, dims, nullptr);
        const hid_t type = h5.Tcopy(*h5.native_float);
        bool ok = space >= 0 && type >= 0 && h5.Tset_order(type, kH5T_ORDER_LE) >= 0;
        hid_t dset = -1;
        if (ok) dset = h5.Dcreate2(file, d.name.c_str(), type, space, kH5P_DEFAULT, kH5P_DEFAULT, kH5P_DEFAULT);
        ok = ok && dset >= 0 && h5.Dwrite(dset, *h5.native_float, kH5S_ALL, kH5S_ALL, kH5P_DEFAULT, d.data) >= 0;
        if (dset >= 0) h5.Dclose(dset);
        if (type >= 0) h5.Tclose(type);
        if (space >= 0) h5.Sclose(space);
        if (!ok) { h5.Fclose(file); NVSM_LOG(FATAL) << "failed to write dataset " << d.name << " to " << filename; }
    }
    if (h5.Fclose(file) < 0) NVSM_LOG(FATAL) << "failed to close " << filename;
}

}  // namespace nvsm_host
