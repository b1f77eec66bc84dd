"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol the
public header declares, and refuses to run without a GPU (no CPU fallback, no oracle in the product path)."""
import ctypes as C
import os
import re

import pytest

import cunvsm_amd as ca
from tests.conftest import ROOT, gpu_available


def test_library_exports_every_declared_symbol():
    ca.build_library()
    L = ca.lib()
    names = ca.abi_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libcunvsm_amd.so does not export %s" % n


def test_unit_test_hooks_live_in_their_own_library():
    """nvsm_debug_* were exported from the PRODUCT library. They are declared in include/cunvsm_amd_test_hooks.h, built
    into libcunvsm_amd_testhooks.so (no kernels of its own: it calls the product's launchers), and libcunvsm_amd.so exports none."""
    import subprocess
    ca.build_library()
    hooks = ca._lib.hook_symbols()
    assert len(hooks) >= 7 and all(h.startswith("nvsm_debug_") for h in hooks)
    assert not set(hooks) & set(ca.abi_symbols())
    exported = subprocess.run(["nm", "-D", "--defined-only", ca.library_path()], capture_output=True, text=True, check=True).stdout
    assert "nvsm_debug_" not in exported and "nvsm_step" in exported
    hooks_so = os.path.join(ROOT, "cunvsm_amd", "libcunvsm_amd_testhooks.so")
    exported = subprocess.run(["nm", "-D", "--defined-only", hooks_so], capture_output=True, text=True, check=True).stdout
    for h in hooks:
        assert " T " + h in exported, h
    L = ca.lib()
    for h in hooks:
        assert hasattr(L, h)          # (the binding finds them through the hooks library)
    with pytest.raises(AttributeError):
        L.product.nvsm_debug_gemm


def test_version_and_defaults():
    L = ca.lib()
    assert b"gfx950" in L.nvsm_version()
    cfg = ca.default_config()
    # scripts/functions.sh:380-399 + NVSM flags (:266)
    assert (cfg.word_repr_size, cfg.entity_repr_size, cfg.window_size) == (300, 256, 10)
    assert cfg.batch_normalization == 1 and cfg.nonlinearity == ca.HARD_TANH and cfg.clip_sigmoid == 1
    assert cfg.update_method == ca.ADAM and cfg.adam_mode == ca.ADAM_DENSE_UPDATE_DENSE_VARIANCE
    assert abs(cfg.regularization_lambda - 1e-2) < 1e-9 and cfg.max_batch_size == 51200


def test_null_arguments_are_status_codes_not_crashes():
    L = ca.lib()
    assert L.nvsm_create(None, None) == 1
    assert L.nvsm_compute_gradients(None) == 1
    assert b"null" in L.nvsm_last_error()


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_bind_host_thread_without_a_device_is_a_status_code():
    import os
    before = os.sched_getaffinity(0)
    with pytest.raises(ca.NvsmError) as e:
        ca.bind_host_thread(0)
    assert e.value.status == 5          # NVSM_ERR_NO_DEVICE
    assert os.sched_getaffinity(0) == before


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    cfg = ca.default_config(num_words=10, num_entities=10, max_batch_size=8)
    with pytest.raises(ca.NvsmError) as e:
        ca.Model(cfg)
    assert e.value.status == 5          # NVSM_ERR_NO_DEVICE


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cunvsm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                with open(os.path.join(dirpath, f), errors="ignore") as fh:
                    src = fh.read()
                assert not re.search(r"nvsm_oracle|from oracle|import oracle|oracle/", src), os.path.join(dirpath, f)


def test_config_struct_matches_header_size():
    # 2 x int64 + 20 x int32/float + 4 reserved int32 = 16 + 24*4 = 112 bytes
    assert C.sizeof(ca.NvsmConfig) == 112
    assert C.sizeof(ca.NvsmBatch) == 48


def test_public_headers_compile(tmp_path):
    """include/cunvsm_amd.h is plain C (a cgo / JNI / ctypes binding can consume it); the C++ wrapper needs C++11 only."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "cunvsm_amd.h")])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", "-I", inc, os.path.join(inc, "cunvsm_amd_test_hooks.h")])
    src = tmp_path / "use.cpp"
    src.write_text('#include "cunvsm_amd/model.hpp"\nint main() { nvsm_config c; nvsm_config_default(&c); return c.device; }\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
