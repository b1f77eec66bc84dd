"""ctypes binding of include/cunvsm_amd.h (the stub a maintainer would write)."""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.environ.get("CUNVSM_AMD_LIB") or os.path.join(_HERE, "libcunvsm_amd.so")      # override: A/B of two builds
_HEADER = os.path.join(_ROOT, "include", "cunvsm_amd.h")
_HOOKS_SO = os.path.join(_HERE, "libcunvsm_amd_testhooks.so")      # nvsm_debug_*: unit-test / profiling hooks, NOT in the product library
_HOOKS_HEADER = os.path.join(_ROOT, "include", "cunvsm_amd_test_hooks.h")

TANH, HARD_TANH = 0, 1
SGD, ADAGRAD, ADAM = 0, 1, 2
ADAM_NONE, ADAM_SPARSE, ADAM_DENSE_UPDATE, ADAM_DENSE_UPDATE_DENSE_VARIANCE = 0, 1, 2, 3
SAMPLER_HOST_MINSTD, SAMPLER_DEVICE = 0, 1

STATUS = {0: "OK", 1: "INVALID_ARGUMENT", 2: "UNSUPPORTED", 3: "DEVICE", 4: "STATE", 5: "NO_DEVICE"}


class NvsmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("nvsm status %d (%s): %s" % (status, STATUS.get(status, "?"), message))
        self.status = status


class NvsmConfig(C.Structure):
    _fields_ = [
        ("num_words", C.c_int64), ("num_entities", C.c_int64),
        ("word_repr_size", C.c_int32), ("entity_repr_size", C.c_int32),
        ("batch_normalization", C.c_int32), ("nonlinearity", C.c_int32),
        ("clip_sigmoid", C.c_int32), ("bias_negative_samples", C.c_int32),
        ("l2_normalize_phrase_reprs", C.c_int32), ("l2_normalize_entity_reprs", C.c_int32),
        ("window_size", C.c_int32), ("num_random_entities", C.c_int32),
        ("regularization_lambda", C.c_float),
        ("update_method", C.c_int32), ("adam_mode", C.c_int32), ("max_batch_size", C.c_int32),
        ("device", C.c_int32), ("sampler", C.c_int32),
        ("world_size", C.c_int32), ("rank", C.c_int32), ("sync_batch_norm", C.c_int32),
        ("dp_exact_tables", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


class NvsmBatch(C.Structure):
    _fields_ = [
        ("features", C.c_void_p), ("feature_weights", C.c_void_p), ("labels", C.c_void_p), ("weights", C.c_void_p),
        ("num_instances", C.c_int64), ("on_device", C.c_int32),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_void_p)


def library_path():
    return _SO


def build_library(force=False):
    """Compiles the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"] + (["-B"] if force else []))
    return _SO


def _declared(header):
    with open(header) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nvsm_[a-z0-9_]+)\s*\(", src)) - {"nvsm_allreduce_fn"})


def abi_symbols():
    """Every function the public header declares (used by the CPU-side ABI test)."""
    return _declared(_HEADER)


def hook_symbols():
    """Every function include/cunvsm_amd_test_hooks.h declares: exported by libcunvsm_amd_testhooks.so, never by the product library."""
    return _declared(_HOOKS_HEADER)


class _Library:
    """The product library; a name it does not export (nvsm_debug_*) is looked up in the test-hooks library, loaded on first use."""

    def __init__(self, product):
        self.__dict__["product"] = product
        self.__dict__["hooks"] = None

    def __getattr__(self, name):
        try:
            return getattr(self.product, name)
        except AttributeError:
            if not name.startswith("nvsm_debug_"):
                raise
        if self.hooks is None:
            if not os.path.exists(_HOOKS_SO):
                raise AttributeError("%s lives in %s, which is missing (make -C cunvsm_amd/csrc)" % (name, _HOOKS_SO))
            H = C.CDLL(_HOOKS_SO, mode=C.RTLD_GLOBAL)      # (its launchers resolve against the product library already loaded)
            vp, i64 = C.c_void_p, C.c_int64
            P = C.POINTER
            for hook, (res, args) in {
                "nvsm_debug_delay": (C.c_int, [vp, C.c_int]),
                "nvsm_debug_set_table_pass_form": (C.c_int, [C.c_int]),
                "nvsm_debug_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
                "nvsm_debug_gemm_time": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_float)]),
                "nvsm_debug_dt_time": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_float), P(C.c_float)]),
                "nvsm_debug_sort": (C.c_int, [i64, C.c_int, vp, vp, vp, C.c_int, P(C.c_float)]),
                "nvsm_debug_gather_mean": (C.c_int, [i64, C.c_int, vp, vp, vp, C.c_int, i64, vp]),
            }.items():
                fn = getattr(H, hook)
                fn.restype = res
                fn.argtypes = args
            self.__dict__["hooks"] = H
        return getattr(self.hooks, name)


_lib = None


def lib():
    """Loads libcunvsm_amd.so; fails loudly when the HIP extension is missing (no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(cunvsm_amd has no CPU fallback)" % _SO)
    try:
        # PyTorch ships its own libamdhip64 / librccl; load it first so one HIP runtime serves both.
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(_SO, mode=C.RTLD_GLOBAL)
    vp, i64, cp = C.c_void_p, C.c_int64, C.c_char_p
    P = C.POINTER
    sig = {
        "nvsm_last_error": (cp, []), "nvsm_version": (cp, []), "nvsm_device_count": (C.c_int, []),
        "nvsm_config_default": (None, [P(NvsmConfig)]),
        "nvsm_create": (C.c_int, [P(NvsmConfig), P(vp)]), "nvsm_destroy": (None, [vp]),
        "nvsm_initialize": (C.c_int, [vp, C.c_uint64]), "nvsm_initialize_from_rng_state": (C.c_int, [vp]),
        "nvsm_host_alloc": (C.c_int, [C.c_size_t, P(vp)]), "nvsm_host_free": (C.c_int, [vp]),
        "nvsm_bind_host_thread": (C.c_int, [C.c_int, P(C.c_int)]),
        "nvsm_comm_selftest": (C.c_int, [C.c_int]),
        "nvsm_comm_latency": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_float), P(i64)]),
        "nvsm_rng_get_state": (C.c_int, [vp, P(C.c_uint64)]), "nvsm_rng_set_state": (C.c_int, [vp, C.c_uint64]),
        "nvsm_param_size": (C.c_int, [vp, cp, P(i64)]),
        "nvsm_get_param": (C.c_int, [vp, cp, vp, i64]), "nvsm_set_param": (C.c_int, [vp, cp, vp, i64]),
        "nvsm_increment_parameter": (C.c_int, [vp, cp, i64, C.c_float]),
        "nvsm_compute_cost": (C.c_int, [vp, P(NvsmBatch), vp]), "nvsm_compute_gradients": (C.c_int, [vp]),
        "nvsm_update": (C.c_int, [vp, C.c_float, C.c_float]), "nvsm_get_cost": (C.c_int, [vp, P(C.c_float)]), "nvsm_get_cost_f64": (C.c_int, [vp, P(C.c_double)]),
        "nvsm_scaled_regularization_lambda": (C.c_float, [vp]),
        "nvsm_step": (C.c_int, [vp, P(NvsmBatch), vp, C.c_float, P(C.c_float)]),
        "nvsm_step_deferred": (C.c_int, [vp, P(NvsmBatch), vp, C.c_float, P(i64)]),
        "nvsm_deferred_cost": (C.c_int, [vp, i64, P(C.c_float)]), "nvsm_wait_inputs": (C.c_int, [vp]),
        "nvsm_tensor_size": (C.c_int, [vp, cp, P(i64)]), "nvsm_get_tensor": (C.c_int, [vp, cp, vp, i64]),
        "nvsm_set_stream": (C.c_int, [vp, vp]), "nvsm_synchronize": (C.c_int, [vp]),
        "nvsm_describe": (C.c_int, [vp, C.c_int64, C.c_char_p, C.c_int64]),
        "nvsm_comm_unique_id": (C.c_int, [vp]), "nvsm_comm_init": (C.c_int, [vp, vp]),
        "nvsm_set_allreduce_callback": (C.c_int, [vp, ALLREDUCE_FN, vp]),
        "nvsm_comm_size": (C.c_int, [vp, P(C.c_int)]), "nvsm_dp_average_tables": (C.c_int, [vp]),
        "nvsm_range_push": (None, [cp]), "nvsm_range_pop": (None, []),
        "nvsm_profile_enable": (C.c_int, [vp, C.c_int]), "nvsm_profile_reset": (C.c_int, [vp]),
        "nvsm_profile_names": (C.c_int, [vp, vp, i64]), "nvsm_profile_select": (C.c_int, [vp, cp]),
        "nvsm_profile_get": (C.c_int, [vp, cp, P(C.c_double), P(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = _Library(L)
    return _lib


def check(status):
    if status != 0:
        raise NvsmError(status, lib().nvsm_last_error().decode())


def device_count():
    return lib().nvsm_device_count()


def bind_host_thread(device=0):
    """nvsm_bind_host_thread: the calling thread onto the CPUs of the device's NUMA node. Returns that node (-1: unknown)."""
    node = C.c_int(-1)
    check(lib().nvsm_bind_host_thread(int(device), C.byref(node)))
    return node.value
