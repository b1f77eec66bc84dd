// Shared by c_api.cpp (libcunvsm_amd.so, the product's C ABI) and test_hooks.cpp (libcunvsm_amd_testhooks.so, the unit-test hooks):
// the handle type and the exception -> status translation. Not part of the ABI.
#pragma once

#include <string>

#include "model.h"

struct nvsm_model {
    cunvsm::Model impl;
    explicit nvsm_model(const nvsm_config& c) : impl(c) {}
};

namespace cunvsm {
void set_last_error(const std::string& what);      // what nvsm_last_error() returns on the calling thread (c_api.cpp)

template <typename Fn>
inline int guarded(Fn&& fn) {
    try {
        fn();
        return NVSM_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.status;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return NVSM_ERR_DEVICE;
    } catch (...) {
        set_last_error("unknown error");
        return NVSM_ERR_DEVICE;
    }
}

// a call on a handle runs under the handle's switches (tuning.h): read once, by nvsm_create
template <typename Fn>
inline int guarded_on(nvsm_model* m, Fn&& fn) {
    TuningScope scope(&m->impl.tune());
    return guarded(fn);
}
// a hook without a handle reads the environment itself, per call: tests switch variables between calls
template <typename Fn>
inline int guarded_hook(Fn&& fn) {
    const Tuning t = Tuning::from_env();
    TuningScope scope(&t);
    return guarded(fn);
}
}  // namespace cunvsm

#define NVSM_REQUIRE(ptr)                                          \
    if (!(ptr)) {                                                  \
        ::cunvsm::set_last_error("null argument: " #ptr);          \
        return NVSM_ERR_INVALID_ARGUMENT;                          \
    }
