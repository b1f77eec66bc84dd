"""cunvsm_amd — MI355X-native NVSM / LSE training hot path.

Python host-side mirror of cuNVSM's ``Model<TextEntity::Objective>`` (include/cuNVSM/model.h:75-131)
over the C ABI of ``libcunvsm_amd.so`` (include/cunvsm_amd.h). There is no CPU fallback: importing
works anywhere (so that the ABI can be inspected), but creating a model without the HIP library or
without a GPU raises.
"""
from ._lib import (  # noqa: F401
    ADAGRAD, ADAM, ADAM_DENSE_UPDATE, ADAM_DENSE_UPDATE_DENSE_VARIANCE, ADAM_NONE, ADAM_SPARSE, HARD_TANH,
    SAMPLER_DEVICE, SAMPLER_HOST_MINSTD, SGD, TANH, NvsmBatch, NvsmConfig, NvsmError, abi_symbols, build_library,
    bind_host_thread, device_count, lib, library_path,
)
from . import dp  # noqa: F401
from .model import Batch, Model, UPDATE_METHODS, default_config  # noqa: F401

__all__ = ["Model", "Batch", "default_config", "UPDATE_METHODS", "NvsmConfig", "NvsmBatch", "NvsmError", "lib",
           "library_path", "build_library", "device_count", "bind_host_thread", "abi_symbols"]
