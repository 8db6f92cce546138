"""ctypes binding of libsedumi_hip.so (include/sedumi_hip.h).

There is exactly one product library: ``sedumi_amd/lib/libsedumi_hip.so`` built by
hipcc for gfx950.  ``lib()`` raises if it is missing -- there is no CPU fallback.
Tests may call ``use_library(path)`` to point the binding at the fiber-emulated
build of the same sources (tests/hipemu); nothing in the package does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libsedumi_hip.so")
_lib = None
_lib_path = None

I64P = C.POINTER(C.c_int64)
F64P = C.POINTER(C.c_double)


class SdmError(RuntimeError):
    pass


class CholPars(C.Structure):
    _fields_ = [("canceltol", C.c_double), ("maxu", C.c_double), ("abstol", C.c_double)]


class Cone(C.Structure):
    _fields_ = [("lpN", C.c_int64), ("lorN", C.c_int64), ("lorNL", I64P), ("sdpN", C.c_int64),
                ("rsdpN", C.c_int64), ("sdpNL", I64P)]


def use_library(path):
    """Select the shared object to bind (tests only).  Resets the cached handle."""
    global _lib, _lib_path
    _lib = None
    _lib_path = path


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _lib_path or DEFAULT_LIB
    if not os.path.exists(path):
        raise SdmError(f"{path} not found: build it with `python -m sedumi_amd.build` "
                       "(hipcc, gfx950).  sedumi_amd has no CPU fallback.")
    L = C.CDLL(path)
    L.sdm_last_error.restype = C.c_char_p
    L.sdm_backend.restype = C.c_char_p
    L.sdm_plan_create.restype = C.c_void_p
    L.sdm_plan_create.argtypes = [C.c_int, C.c_void_p]
    L.sdm_plan_destroy.argtypes = [C.c_void_p]
    L.sdm_plan_devptr.restype = C.c_void_p
    L.sdm_plan_devptr.argtypes = [C.c_void_p, C.c_char_p, I64P]
    for name in ("sdm_plan_sync", "sdm_plan_getada", "sdm_plan_getdatq", "sdm_plan_fwsolve", "sdm_plan_bwsolve", "sdm_plan_ldlsolve"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.sdm_plan_blkchol.argtypes = [C.c_void_p, C.POINTER(CholPars), C.c_int]
    L.sdm_plan_upload.argtypes = [C.c_void_p, C.c_char_p, F64P, C.c_int64]
    L.sdm_plan_download.argtypes = [C.c_void_p, C.c_char_p, F64P, C.c_int64]
    L.sdm_plan_timer_begin.argtypes = [C.c_void_p, C.c_int]
    L.sdm_plan_timer_end.argtypes = [C.c_void_p, C.c_int]
    L.sdm_plan_timer_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    _lib = L
    return L


def check(rc):
    if rc:
        raise SdmError(lib().sdm_last_error().decode())


def backend():
    return lib().sdm_backend().decode()


def device_count():
    return int(lib().sdm_device_count())


# ---- numpy marshalling helpers -------------------------------------------------
def i64(a):
    return np.ascontiguousarray(np.asarray(a).ravel(), dtype=np.int64)


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))


def pi(a):
    return a.ctypes.data_as(I64P) if a is not None else None


def pf(a):
    return a.ctypes.data_as(F64P) if a is not None else None


def make_cone(lpN, q, s, rsdpN=None):
    """Returns (Cone struct, keep-alive tuple)."""
    qa, sa = i64(q), i64(s)
    K = Cone(int(lpN), len(qa), pi(qa) if len(qa) else None, len(sa),
             int(len(sa) if rsdpN is None else rsdpN), pi(sa) if len(sa) else None)
    return K, (qa, sa)
