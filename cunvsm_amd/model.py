"""Host-side mirror of ``Model<TextEntity::Objective>`` and ``TextEntity::Batch`` over the C ABI.

Same method names and argument meaning as the reference (include/cuNVSM/model.h:97-115):
``initialize``, ``compute_cost``, ``compute_gradients``, ``update``, ``get_cost``; the forward result
and the gradients live inside the handle (the reference returns owning pointers, cpp/main.cu:405-411).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import NvsmBatch, NvsmConfig, check, lib

# --update_method of the reference CLI (cpp/main.cu:479-485)
UPDATE_METHODS = {
    "sgd": (_lib.SGD, _lib.ADAM_NONE),
    "adagrad": (_lib.ADAGRAD, _lib.ADAM_NONE),
    "sparse_adam": (_lib.ADAM, _lib.ADAM_SPARSE),
    "dense_adam": (_lib.ADAM, _lib.ADAM_DENSE_UPDATE),
    "full_adam": (_lib.ADAM, _lib.ADAM_DENSE_UPDATE_DENSE_VARIANCE),
}
NONLINEARITIES = {"tanh": _lib.TANH, "hard_tanh": _lib.HARD_TANH}      # cpp/main.cu:487-490

PARAM_NAMES = (
    "word_representations-representations",
    "entity_representations-representations",
    "word_entity_mapping-transform",
    "word_entity_mapping-bias",
)


def default_config(**overrides):
    """nvsm_config with the reference CLI defaults, then keyword overrides.
    ``update_method`` / ``nonlinearity`` accept the CLI strings."""
    cfg = NvsmConfig()
    lib().nvsm_config_default(C.byref(cfg))
    for k, v in overrides.items():
        if k == "update_method" and isinstance(v, str):
            cfg.update_method, cfg.adam_mode = UPDATE_METHODS[v]
        elif k == "nonlinearity" and isinstance(v, str):
            cfg.nonlinearity = NONLINEARITIES[v]
        else:
            if not hasattr(cfg, k):
                raise AttributeError("nvsm_config has no field %r" % k)
            setattr(cfg, k, int(v) if isinstance(v, (bool, np.bool_)) else v)
    return cfg


class PinnedArray:
    """A numpy array over page-locked host memory from nvsm_host_alloc — what the reference's Batch allocates with
    cudaHostAlloc (cpp/data.cu:8-40): host→device copies from it are asynchronous. `.array` is the numpy view."""

    def __init__(self, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        self._ptr = C.c_void_p()
        check(lib().nvsm_host_alloc(max(1, n * dtype.itemsize), C.byref(self._ptr)))
        buf = (C.c_char * (n * dtype.itemsize)).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

    def __del__(self):
        try:
            if self._ptr:
                lib().nvsm_host_free(self._ptr)
                self._ptr = None
        except Exception:   # pragma: no cover  (interpreter shutdown)
            pass


def pinned_copy(a):
    """Copy of `a` in page-locked memory; keep the returned PinnedArray alive as long as its .array is in use."""
    a = np.asarray(a)
    p = PinnedArray(a.shape, a.dtype)
    p.array[...] = a
    return p


class Batch:
    """TextEntity::Batch (include/cuNVSM/data.h:114-177): features [B*w] int64, feature_weights [B*w] float,
    labels [B] int64, weights [B] float. Arrays may be numpy (host) or torch CUDA tensors (already in HBM)."""

    def __init__(self, features, labels, feature_weights=None, weights=None):
        self.features, self.labels = features, labels
        self.feature_weights, self.weights = feature_weights, weights
        self.on_device = hasattr(features, "data_ptr")
        if self.on_device:
            import torch
            assert features.dtype == torch.int64 and labels.dtype == torch.int64
            for t in (feature_weights, weights):
                assert t is None or t.dtype == torch.float32
            self.num_instances = int(labels.numel())
        else:
            self.features = np.ascontiguousarray(features, dtype=np.int64)
            self.labels = np.ascontiguousarray(labels, dtype=np.int64)
            if feature_weights is not None:
                self.feature_weights = np.ascontiguousarray(feature_weights, dtype=np.float32)
            if weights is not None:
                self.weights = np.ascontiguousarray(weights, dtype=np.float32)
            self.num_instances = int(self.labels.size)

    def _ptr(self, a):
        if a is None:
            return None
        return a.data_ptr() if self.on_device else a.ctypes.data

    def check_shapes(self, window_size):
        """The ABI takes bare pointers: the element counts it will read are asserted here (include/cunvsm_amd.h,
        index contract). Id RANGES are checked on the device."""
        size = (lambda a: int(a.numel())) if self.on_device else (lambda a: int(a.size))
        B = self.num_instances
        if size(self.features) != B * window_size:
            raise ValueError("features holds %d ids, expected num_instances * window_size = %d" % (size(self.features), B * window_size))
        if self.feature_weights is not None and size(self.feature_weights) != B * window_size:
            raise ValueError("feature_weights holds %d values, expected %d" % (size(self.feature_weights), B * window_size))
        if self.weights is not None and size(self.weights) != B:
            raise ValueError("weights holds %d values, expected %d" % (size(self.weights), B))

    def as_struct(self):
        return NvsmBatch(self._ptr(self.features), self._ptr(self.feature_weights), self._ptr(self.labels),
                         self._ptr(self.weights), self.num_instances, int(self.on_device))


class Model:
    def __init__(self, cfg):
        self.cfg = cfg
        self._h = C.c_void_p()
        check(lib().nvsm_create(C.byref(cfg), C.byref(self._h)))
        self._cb = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().nvsm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- Model::initialize(RNG*) ------------------------------------------------------------------
    def initialize(self, seed):
        check(lib().nvsm_initialize(self._h, seed))

    @property
    def rng_state(self):
        s = C.c_uint64()
        check(lib().nvsm_rng_get_state(self._h, C.byref(s)))
        return s.value

    @rng_state.setter
    def rng_state(self, s):
        check(lib().nvsm_rng_set_state(self._h, s))

    # -- the step ----------------------------------------------------------------------------------
    def compute_cost(self, batch, entity_ids=None):
        ids = None
        if entity_ids is not None:
            ids = np.ascontiguousarray(entity_ids, dtype=np.int64)
        st = self._checked(batch, ids)
        self._keep = (batch, ids)
        check(lib().nvsm_compute_cost(self._h, C.byref(st), None if ids is None else ids.ctypes.data))

    def _checked(self, batch, ids):
        batch.check_shapes(self.cfg.window_size)
        if ids is not None and ids.size != batch.num_instances * (self.cfg.num_random_entities + 1):
            raise ValueError("entity_ids holds %d ids, expected num_instances * (num_random_entities + 1) = %d"
                             % (ids.size, batch.num_instances * (self.cfg.num_random_entities + 1)))
        return batch.as_struct()

    def compute_gradients(self):
        check(lib().nvsm_compute_gradients(self._h))

    def update(self, learning_rate, scaled_regularization_lambda=None):
        if scaled_regularization_lambda is None:
            scaled_regularization_lambda = self.scaled_regularization_lambda()
        check(lib().nvsm_update(self._h, learning_rate, scaled_regularization_lambda))

    def get_cost(self):
        c = C.c_float()
        check(lib().nvsm_get_cost(self._h, C.byref(c)))
        return c.value

    def scaled_regularization_lambda(self):
        return lib().nvsm_scaled_regularization_lambda(self._h)

    def step(self, batch, learning_rate, entity_ids=None, want_cost=False):
        ids = None
        if entity_ids is not None:
            ids = np.ascontiguousarray(entity_ids, dtype=np.int64)
        st = self._checked(batch, ids)
        self._keep = (batch, ids)
        c = C.c_float()
        check(lib().nvsm_step(self._h, C.byref(st), None if ids is None else ids.ctypes.data, learning_rate,
                              C.byref(c) if want_cost else None))
        return c.value if want_cost else None

    # -- parameters / tensors ----------------------------------------------------------------------
    def step_deferred(self, batch, learning_rate, entity_ids=None):
        """nvsm_step + an asynchronous read-back of its loss; returns the ticket for deferred_cost()."""
        ids = None
        if entity_ids is not None:
            ids = np.ascontiguousarray(entity_ids, dtype=np.int64)
        st = self._checked(batch, ids)
        self._keep = (batch, ids)
        t = C.c_int64()
        check(lib().nvsm_step_deferred(self._h, C.byref(st), ids.ctypes.data if ids is not None else None, float(learning_rate), C.byref(t)))
        return t.value

    def deferred_cost(self, ticket):
        c = C.c_float()
        check(lib().nvsm_deferred_cost(self._h, int(ticket), C.byref(c)))
        return c.value

    def wait_inputs(self):
        check(lib().nvsm_wait_inputs(self._h))

    def get_param(self, name):
        n = C.c_int64()
        check(lib().nvsm_param_size(self._h, name.encode(), C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        check(lib().nvsm_get_param(self._h, name.encode(), out.ctypes.data, n.value))
        return out

    def set_param(self, name, value):
        v = np.ascontiguousarray(value, dtype=np.float32).ravel()
        check(lib().nvsm_set_param(self._h, name.encode(), v.ctypes.data, v.size))

    def get_data(self):
        """ModelBase::get_data() (cpp/model.cu:64-93): the four tensors the HDF5 writer dumps."""
        return {n: self.get_param(n) for n in PARAM_NAMES}

    def get_tensor(self, name):
        n = C.c_int64()
        check(lib().nvsm_tensor_size(self._h, name.encode(), C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        check(lib().nvsm_get_tensor(self._h, name.encode(), out.ctypes.data, n.value))
        return out

    # -- runtime -----------------------------------------------------------------------------------
    def synchronize(self):
        check(lib().nvsm_synchronize(self._h))

    def set_stream(self, stream_ptr):
        check(lib().nvsm_set_stream(self._h, stream_ptr))

    def comm_init(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(lib().nvsm_comm_init(self._h, buf))

    def describe(self, batch=None):
        """Which kernels a step of `batch` windows takes on this handle, table modes, switches off their defaults (nvsm_describe)."""
        buf = C.create_string_buffer(2048)
        check(lib().nvsm_describe(self._h, int(batch if batch is not None else self.cfg.max_batch_size), buf, 2048))
        return buf.value.decode()

    def comm_size(self):
        """Ranks of the engine's RCCL communicator as ncclCommCount reports them (0: none built)."""
        n = C.c_int()
        check(lib().nvsm_comm_size(self._h, C.byref(n)))
        return n.value

    def dp_average_tables(self):
        """Collective: word / document tables ← mean over the ranks' replicas (see include/cunvsm_amd.h)."""
        check(lib().nvsm_dp_average_tables(self._h))

    def set_allreduce_callback(self, fn):
        """fn(numpy float64 array) must sum the array in place across ranks."""
        def tramp(ptr, n, _user):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(n,)))
                return 0
            except Exception:  # pragma: no cover
                return 1
        self._cb = _lib.ALLREDUCE_FN(tramp)
        check(lib().nvsm_set_allreduce_callback(self._h, self._cb, None))

    def profile_enable(self, on=True):
        check(lib().nvsm_profile_enable(self._h, int(on)))

    def profile_select(self, kernel=None):
        """Time only this kernel group (None = all of them)."""
        check(lib().nvsm_profile_select(self._h, kernel.encode() if kernel else None))

    def debug_delay(self, microseconds):
        """Spin kernel on the model's stream (profiling aid: lets the host queue a whole step ahead of the GPU)."""
        check(lib().nvsm_debug_delay(self._h, int(microseconds)))

    def profile_reset(self):
        check(lib().nvsm_profile_reset(self._h))

    def profile(self):
        """{kernel name: (total_ms, launches)} measured with HIP events on the model's stream."""
        buf = C.create_string_buffer(1 << 14)
        check(lib().nvsm_profile_names(self._h, buf, len(buf)))
        names = [n.decode() for n in buf.raw.split(b"\0\0")[0].split(b"\0") if n]
        out = {}
        for n in names:
            ms, cnt = C.c_double(), C.c_int64()
            check(lib().nvsm_profile_get(self._h, n.encode(), C.byref(ms), C.byref(cnt)))
            out[n] = (ms.value, cnt.value)
        return out


def comm_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().nvsm_comm_unique_id(buf))
    return buf.raw
