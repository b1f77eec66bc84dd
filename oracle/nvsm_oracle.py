"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes binding of the CPU oracle (oracle/nvsm_oracle.hpp). Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package cunvsm_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnvsm_oracle.so")

SGD, ADAGRAD, ADAM = 0, 1, 2
ADAM_NONE, ADAM_SPARSE, ADAM_DENSE_UPDATE, ADAM_DENSE_UPDATE_DENSE_VARIANCE = 0, 1, 2, 3
TANH, HARD_TANH = 0, 1
F64, F32 = 0, 1

PARAM_NAMES = (
    "word_representations-representations",
    "entity_representations-representations",
    "word_entity_mapping-transform",
    "word_entity_mapping-bias",
)


def build(force=False):
    # `make` every time: a no-op when the library is newer than its sources, a rebuild when a source changed
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(_SO):
            raise
    return _SO


class OrcConfig(C.Structure):
    _fields_ = [
        ("num_words", C.c_int64), ("num_entities", C.c_int64),
        ("word_dim", C.c_int32), ("entity_dim", C.c_int32), ("window", C.c_int32), ("num_random", C.c_int32),
        ("batch_norm", C.c_int32), ("nonlinearity", C.c_int32), ("clip_sigmoid", C.c_int32),
        ("bias_negative_samples", C.c_int32), ("l2_phrase", C.c_int32), ("l2_entity", C.c_int32),
        ("update_method", C.c_int32), ("adam_mode", C.c_int32),
        ("lambda_", C.c_double), ("bn_epsilon", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
        ("opt_epsilon", C.c_double),
    ]


def effective_cpus():
    """CPUs this process can actually use: the affinity mask capped by the cgroup CPU quota. (The GPU box shows 256
    hardware threads and grants 16 CPUs of time: an OpenMP team of 256 then runs a step 17x slower than a team of 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # v1
                quota, period = int(f.read()), int(g.read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    try:
        L = C.CDLL(_SO)
    except OSError:
        build(force=True)
        L = C.CDLL(_SO)
    vp, i64, dbl, cp = C.c_void_p, C.c_int64, C.c_double, C.c_char_p
    P = C.POINTER
    sig = {
        "orc_rng_create": (vp, [C.c_uint64]), "orc_rng_free": (None, [vp]), "orc_rng_seed": (None, [vp, C.c_uint64]),
        "orc_rng_get_state": (C.c_uint64, [vp]), "orc_rng_set_state": (None, [vp, C.c_uint64]),
        "orc_generate_labels": (None, [vp, vp, i64, i64, i64, vp]),
        "orc_glorot": (None, [vp, C.c_int, i64, i64, vp]),
        "orc_model_create": (vp, [P(OrcConfig), C.c_int]), "orc_model_free": (None, [vp]),
        "orc_model_initialize": (None, [vp, vp]),
        "orc_model_tensor_size": (i64, [vp, cp]), "orc_model_get": (C.c_int, [vp, cp, vp]),
        "orc_model_set": (C.c_int, [vp, cp, vp]), "orc_model_tensor_ptr": (vp, [vp, cp]),
        "orc_model_forward": (None, [vp, vp, vp, vp, vp, i64]),
        "orc_model_forward_native": (None, [vp, vp, vp, vp, vp, i64]),
        "orc_model_get_cost": (dbl, [vp]), "orc_model_backward": (None, [vp]),
        "orc_model_update": (C.c_int, [vp, dbl, dbl]), "orc_model_scaled_lambda": (dbl, [vp]),
        "orc_model_set_allreduce": (None, [vp, ALLREDUCE_FN, vp, C.c_int]),
        "orc_model_set_exact_tables": (None, [vp, C.c_int]),
        "orc_model_set_owner_rows": (None, [vp, C.c_int]),
        "orc_model_gradcheck": (C.c_int, [vp, vp, vp, vp, vp, i64, dbl, dbl, P(dbl), P(C.c_int)]),
        "orc_num_threads": (C.c_int, []), "orc_set_num_threads": (None, [C.c_int]),
        "orc_reps_create": (vp, [i64, i64, C.c_int, C.c_int, dbl, dbl, dbl, C.c_int]), "orc_reps_free": (None, [vp]),
        "orc_reps_fill": (None, [vp, dbl]), "orc_reps_set": (None, [vp, vp]), "orc_reps_get": (i64, [vp, C.c_int, vp]),
        "orc_reps_update": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, dbl, dbl]),
        "orc_reps_update_dense_const": (None, [vp, dbl, dbl, dbl]),
        "orc_tr_create": (vp, [i64, i64, C.c_int, dbl, dbl, dbl, C.c_int]), "orc_tr_free": (None, [vp]),
        "orc_tr_fill": (None, [vp, dbl]), "orc_tr_get": (i64, [vp, C.c_int, vp]),
        "orc_tr_update": (None, [vp, vp, vp, dbl, dbl]),
        "orc_average_repr": (None, [vp, i64, vp, vp, i64, i64, vp]),
        "orc_bn_forward": (None, [vp, i64, i64, vp, dbl, vp, vp, vp]),
        "orc_bn_backward": (None, [vp, vp, i64, i64, vp, vp, vp, vp]),
        "orc_normalizer_forward": (None, [vp, i64, i64, vp, vp]),
        "orc_normalizer_backward": (None, [vp, vp, vp, i64, i64, vp]),
        "orc_truncated_sigmoid": (dbl, [dbl, dbl]), "orc_truncated_sigmoid_f32": (dbl, [dbl, dbl]),
        "orc_sigmoid_deriv": (dbl, [dbl, dbl]),
        "orc_clip": (dbl, [dbl, C.c_int]), "orc_clip_deriv": (dbl, [dbl, C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    set_num_threads(0)
    return L


def set_num_threads(n):
    """OpenMP team of the oracle: n threads, capped by effective_cpus(); n <= 0 = every CPU the process may use."""
    cap = effective_cpus()
    lib().orc_set_num_threads(cap if n <= 0 else min(int(n), cap))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class Rng:
    """std::minstd_rand0 (include/cuNVSM/base.h:36)."""

    def __init__(self, seed=1):
        self.h = lib().orc_rng_create(seed)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_rng_free(self.h)
            self.h = None

    def seed(self, s):
        lib().orc_rng_seed(self.h, s)

    @property
    def state(self):
        return lib().orc_rng_get_state(self.h)

    @state.setter
    def state(self, s):
        lib().orc_rng_set_state(self.h, s)

    def generate_labels(self, labels, num_entities, num_negative):
        """cpp/labels.cu:4-22."""
        labels = _i64(labels)
        out = np.empty(labels.size * (num_negative + 1), dtype=np.int64)
        lib().orc_generate_labels(self.h, _p(labels), num_entities, labels.size, num_negative, _p(out))
        return out

    def glorot(self, rows, cols, dtype=F64):
        out = np.empty(rows * cols, dtype=np.float64)
        lib().orc_glorot(self.h, dtype, rows, cols, _p(out))
        return out


def make_config(num_words, num_entities, word_dim, entity_dim, window, num_random, *, batch_norm=False,
                nonlinearity=TANH, clip_sigmoid=False, bias_negative_samples=False, l2_phrase=False,
                l2_entity=False, lambda_=0.0, update_method=SGD, adam_mode=ADAM_NONE, bn_epsilon=1e-4,
                beta1=0.9, beta2=0.999, opt_epsilon=1e-6):
    return OrcConfig(num_words, num_entities, word_dim, entity_dim, window, num_random, int(batch_norm),
                     nonlinearity, int(clip_sigmoid), int(bias_negative_samples), int(l2_phrase), int(l2_entity),
                     update_method, adam_mode, lambda_, bn_epsilon, beta1, beta2, opt_epsilon)


class Model:
    """Mirror of Model<TextEntity::Objective> (include/cuNVSM/model.h:75-131) on the CPU oracle."""

    def __init__(self, cfg, dtype=F64):
        self.cfg = cfg
        self.dtype = dtype
        self.h = lib().orc_model_create(C.byref(cfg), dtype)
        if not self.h:
            raise RuntimeError("orc_model_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_model_free(self.h)
            self.h = None

    def initialize(self, rng):
        lib().orc_model_initialize(self.h, rng.h)

    def get(self, name):
        n = lib().orc_model_tensor_size(self.h, name.encode())
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, dtype=np.float64)
        lib().orc_model_get(self.h, name.encode(), _p(out))
        return out

    def set(self, name, value):
        v = _f64(value).ravel()
        n = lib().orc_model_tensor_size(self.h, name.encode())
        if n != v.size:
            raise ValueError("%s: expected %d values, got %d" % (name, n, v.size))
        lib().orc_model_set(self.h, name.encode(), _p(v))

    def native_view(self, name):
        """numpy view (model dtype) straight onto the oracle's buffer."""
        n = lib().orc_model_tensor_size(self.h, name.encode())
        ptr = lib().orc_model_tensor_ptr(self.h, name.encode())
        ct = C.c_double if self.dtype == F64 else C.c_float
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))

    def forward(self, words, word_weights, entity_ids, inst_weights):
        words, entity_ids = _i64(words), _i64(entity_ids)
        ww, iw = _f64(word_weights), _f64(inst_weights)
        B = iw.size
        assert words.size == B * self.cfg.window and entity_ids.size == B * (self.cfg.num_random + 1)
        lib().orc_model_forward(self.h, _p(words), _p(ww), _p(entity_ids), _p(iw), B)

    def forward_native(self, words, word_weights, entity_ids, inst_weights):
        B = inst_weights.size
        lib().orc_model_forward_native(self.h, _p(words), _p(word_weights), _p(entity_ids), _p(inst_weights), B)

    def get_cost(self):
        return lib().orc_model_get_cost(self.h)

    def set_allreduce(self, fn, world):
        """Data-parallel test hook: fn(numpy float64 view) sums in place across `world` ranks."""
        def tramp(ptr, n, _user):
            fn(np.ctypeslib.as_array(ptr, shape=(n,)))
            return 0
        self._cb = ALLREDUCE_FN(tramp)
        lib().orc_model_set_allreduce(self.h, self._cb, None, world)

    def set_exact_tables(self, rank):
        """Data-parallel test hook (after set_allreduce): update() applies every rank's sparse gradients, in rank order —
        the single-process update on the global batch. rank < 0 switches it off."""
        lib().orc_model_set_exact_tables(self.h, rank)

    def set_owner_rows(self, on=True):
        """Data-parallel test hook (after set_exact_tables): the documents table partitioned by owner, row r -> rank r mod world:
        a rank applies only the entries of its own rows and the ranks then exchange their rows."""
        lib().orc_model_set_owner_rows(self.h, int(bool(on)))

    def backward(self):
        lib().orc_model_backward(self.h)

    def update(self, lr, scaled_lambda=None):
        if scaled_lambda is None:
            scaled_lambda = self.scaled_regularization_lambda()
        if lib().orc_model_update(self.h, lr, scaled_lambda) != 0:
            raise RuntimeError("oracle update failed")

    def scaled_regularization_lambda(self):
        return lib().orc_model_scaled_lambda(self.h)

    def gradcheck(self, words, word_weights, entity_ids, inst_weights, eps=1e-5, thresh=1e-4):
        words, entity_ids = _i64(words), _i64(entity_ids)
        ww, iw = _f64(word_weights), _f64(inst_weights)
        worst, checked = C.c_double(0), C.c_int(0)
        failed = lib().orc_model_gradcheck(self.h, _p(words), _p(ww), _p(entity_ids), _p(iw), iw.size, eps, thresh,
                                           C.byref(worst), C.byref(checked))
        return failed, checked.value, worst.value


class Reps:
    """RepresentationsStorage + updater (cpp/storage.cu, cpp/updates*.cu) for the unit KATs."""

    def __init__(self, n, dim, method=SGD, mode=ADAM_NONE, beta1=0.9, beta2=0.999, eps=1e-6, dtype=F64):
        self.n, self.dim = n, dim
        self.h = lib().orc_reps_create(n, dim, method, mode, beta1, beta2, eps, dtype)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_reps_free(self.h)
            self.h = None

    def fill(self, v):
        lib().orc_reps_fill(self.h, v)

    def set(self, data):
        lib().orc_reps_set(self.h, _p(_f64(data)))

    def get(self, which=0):
        n = lib().orc_reps_get(self.h, which, None)
        out = np.empty(n, dtype=np.float64)
        lib().orc_reps_get(self.h, which, _p(out))
        return out

    def update(self, groups, lr, lambda_):
        """groups: list of (grad [num_grads*dim], indices, window, weights-or-None). Returns the (modified) grads."""
        k = len(groups)
        grads = [_f64(g[0]).copy() for g in groups]
        idx = [_i64(g[1]) for g in groups]
        wts = [None if g[3] is None else _f64(g[3]) for g in groups]
        gp = (C.c_void_p * k)(*[a.ctypes.data for a in grads])
        ip = (C.c_void_p * k)(*[a.ctypes.data for a in idx])
        wp = (C.c_void_p * k)(*[None if a is None else a.ctypes.data for a in wts])
        ng = (C.c_int64 * k)(*[grads[i].size // self.dim for i in range(k)])
        win = (C.c_int64 * k)(*[g[2] for g in groups])
        rc = lib().orc_reps_update(self.h, k, gp, ng, ip, win, wp, lr, lambda_)
        if rc != 0:
            raise RuntimeError("oracle reps update failed")
        return grads

    def update_dense_const(self, g, lr, lambda_):
        lib().orc_reps_update_dense_const(self.h, g, lr, lambda_)


class Transform:
    def __init__(self, word_dim, entity_dim, method=SGD, beta1=0.9, beta2=0.999, eps=1e-6, dtype=F64):
        self.h = lib().orc_tr_create(word_dim, entity_dim, method, beta1, beta2, eps, dtype)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tr_free(self.h)
            self.h = None

    def fill(self, v):
        lib().orc_tr_fill(self.h, v)

    def get(self, which):
        n = lib().orc_tr_get(self.h, which, None)
        out = np.empty(n, dtype=np.float64)
        lib().orc_tr_get(self.h, which, _p(out))
        return out

    def update(self, grad_transform, grad_bias, lr, lambda_):
        gt, gb = _f64(grad_transform).copy(), _f64(grad_bias).copy()
        lib().orc_tr_update(self.h, _p(gt), _p(gb), lr, lambda_)
        return gt, gb


def average_repr(repr_, dim, indices, weights, window):
    repr_, indices = _f64(repr_), _i64(indices)
    w = None if weights is None else _f64(weights)
    n = indices.size // window
    out = np.empty(n * dim, dtype=np.float64)
    lib().orc_average_repr(_p(repr_), dim, _p(indices), _p(w), n, window, _p(out))
    return out


def bn_forward(x, n, dim, bias, eps):
    x, bias = _f64(x), _f64(bias)
    y, mean, inv = np.empty(n * dim), np.empty(dim), np.empty(dim)
    lib().orc_bn_forward(_p(x), n, dim, _p(bias), eps, _p(y), _p(mean), _p(inv))
    return y, mean, inv


def bn_backward(dy, x, n, dim, mean, inv_std):
    dy, x, mean, inv_std = _f64(dy), _f64(x), _f64(mean), _f64(inv_std)
    dx, gb = np.empty(n * dim), np.empty(dim)
    lib().orc_bn_backward(_p(dy), _p(x), n, dim, _p(mean), _p(inv_std), _p(dx), _p(gb))
    return dx, gb


def normalizer_forward(x, n, dim):
    x = _f64(x)
    y, norms = np.empty(n * dim), np.empty(n)
    lib().orc_normalizer_forward(_p(x), n, dim, _p(y), _p(norms))
    return y, norms


def normalizer_backward(g, x, norms, n, dim):
    g, x, norms = _f64(g), _f64(x), _f64(norms)
    out = np.empty(n * dim)
    lib().orc_normalizer_backward(_p(g), _p(x), _p(norms), n, dim, _p(out))
    return out


def truncated_sigmoid(x, eps, dtype=F64):
    return lib().orc_truncated_sigmoid(x, eps) if dtype == F64 else lib().orc_truncated_sigmoid_f32(x, eps)
