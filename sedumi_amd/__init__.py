"""sedumi_amd -- MI355X (gfx950) implementation of SeDuMi's normal-equations hot path.

Layout
  csrc/      hand-written HIP kernels + the C ABI (include/sedumi_hip.h)
  lib/       libsedumi_hip.so (built in-tree by ``python -m sedumi_amd.build``)
  capi.py    ctypes binding of the C ABI (fails loudly when the library or a GPU is missing)
  mex.py     host-side mirror of the reference MEX interface (same names / argument meaning)
  plan.py    resident-plan wrapper (data stays in HBM across the calls of an IPM iteration)
  problem.py synthetic problem builders in SeDuMi's internal (post-pretransfo) form
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
