"""ctypes bindings for the native libraries.

``cpu`` is always available (built on demand with g++).  ``hip`` is the gfx950 kernel
library; it is required whenever a CUDA/HIP tensor reaches an op in ``evox_amd.ops`` and
loading it never falls back silently: a missing or unloadable library raises.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

from . import build as _build

_lock = threading.Lock()
_cpu = None
_hip = None

P = c_void_p
I64 = c_int64
F = c_float
I = c_int


def _declare(lib, table):
    for name, (restype, argtypes) in table.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes


_CPU_API = {
    "evox_cpu_random_bits": (None, [P, P, I64, I64]),
    "evox_cpu_uniform": (None, [P, P, I64, I64, F, F]),
    "evox_cpu_normal": (None, [P, P, I64, I64, F, F]),
    "evox_cpu_randint": (None, [P, P, I64, I64, I64, I64]),
    "evox_cpu_split": (None, [P, P, I64, I64]),
    "evox_cpu_fold_in": (None, [P, P, I64, I64]),
}

# name -> (restype, argtypes); every HIP entry point returns a hipError_t as int and takes
# the stream as its last argument
_HIP_API = {
    "evox_hip_random_bits": (I, [P, P, I64, I64, P]),
    "evox_hip_uniform": (I, [P, P, I64, I64, F, F, P]),
    "evox_hip_normal": (I, [P, P, I64, I64, F, F, P]),
    "evox_hip_normal_bf16": (I, [P, P, I64, I64, F, F, P]),
    "evox_hip_randint": (I, [P, P, I64, I64, I64, I64, P]),
    "evox_hip_split": (I, [P, P, I64, I64, P]),
    "evox_hip_split_nested": (I, [P, P, I, I, I, P]),
    "evox_hip_fold_in": (I, [P, P, I64, I64, P]),
}


def register_hip_api(table):
    """Let op modules declare the entry points they own (keeps this file small)."""
    _HIP_API.update(table)
    if _hip is not None:
        _declare(_hip, table)


def register_cpu_api(table):
    _CPU_API.update(table)
    if _cpu is not None:
        _declare(_cpu, table)


def cpu():
    global _cpu
    if _cpu is None:
        with _lock:
            if _cpu is None:
                path = _build.build_cpu()
                lib = ctypes.CDLL(str(path))
                _declare(lib, _CPU_API)
                _cpu = lib
    return _cpu


def hip():
    global _hip
    if _hip is None:
        with _lock:
            if _hip is None:
                path = _build.HIP_LIB
                if _build.have_hipcc() and os.environ.get("EVOX_AMD_NO_AUTOBUILD") != "1":
                    path = _build.build_hip()
                if not path.exists():
                    raise RuntimeError(
                        f"{path} is missing: run `python -m evox_amd._native.build` (needs hipcc) "
                        "before using evox_amd on a GPU"
                    )
                lib = ctypes.CDLL(str(path))
                _declare(lib, _HIP_API)
                _hip = lib
    return _hip


def hip_loaded() -> bool:
    return _hip is not None


def check(status: int, what: str = "kernel launch"):
    if status != 0:
        raise RuntimeError(f"HIP error {status} in {what}")
